"""Drop-in for the reference's `models` package (put `neuraludf_amd/` on sys.path and
`from models.fields import ...` resolves here)."""
