"""Container for ciphertexts, plaintext polynomials and keys.

Field-for-field the reference's NamedTuple (src/liberate/fhe/data_struct.py:5-24): `data` holds one
int64 tensor [rows, N] per participating GPU (or nested data_structs for key-switch / galois keys),
the flags say which representation the rows are in.
"""
from typing import NamedTuple

from .version import VERSION


class data_struct(NamedTuple):
    data: tuple | list
    include_special: bool
    ntt_state: bool
    montgomery_state: bool
    origin: str
    level: int
    hash: str
    version: str = VERSION
