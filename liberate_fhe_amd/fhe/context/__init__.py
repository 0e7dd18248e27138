from .ckks_context import ckks_context
