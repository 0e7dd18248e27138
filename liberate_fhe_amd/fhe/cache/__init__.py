from . import cache
