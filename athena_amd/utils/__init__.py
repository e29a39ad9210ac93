"""Host-side utilities (mirror of the reference's athena/utils)."""
