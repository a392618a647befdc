"""placeholder: mesh export is outside the hot path (the launcher never reaches it)."""
