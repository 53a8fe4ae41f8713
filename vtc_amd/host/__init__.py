"""Host-side mirror of the reference's operator/plugin interface for the hot path."""
