"""MI355X-native RNS-CKKS ciphertext arithmetic behind the liberate.fhe API."""
from . import fhe, ntt, csprng
from .csprng import Csprng
