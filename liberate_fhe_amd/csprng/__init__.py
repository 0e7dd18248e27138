from .csprng import Csprng
