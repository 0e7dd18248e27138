# API level mirrored from the reference (src/liberate/fhe/version.py); data_structs carry it.
VERSION: str = "v0.9.0"
