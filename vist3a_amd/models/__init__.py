"""Host-side mirror of the reference's module interface for the text->3DGS path."""
