from . import context, encdec, presets
from .ckks_engine import ckks_engine
from .data_struct import data_struct
from .presets import params
