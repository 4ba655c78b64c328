"""Stitched Conv3d -> AnySplat reconstruction (engine + weight helpers)."""
