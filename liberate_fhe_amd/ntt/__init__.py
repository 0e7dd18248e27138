from . import ntt_cuda
from .ntt_context import ntt_context
