"""placeholder: h5py is imported by the reference runner but unused on the training path."""
