"""MI355X kernels of the hot path: ctypes binding (`_lib`) + autograd wrappers (`ops`)."""
