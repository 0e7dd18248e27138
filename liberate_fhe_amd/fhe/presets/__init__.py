from .params import params
from . import types
from . import errors
