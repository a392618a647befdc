"""Drop-in for the reference's `loss` package."""
