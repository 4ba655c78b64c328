"""vist3a_amd — MI355X (gfx950) native implementation of the VIST3A text->3DGS inference hot path."""
__version__ = "0.1.0"
