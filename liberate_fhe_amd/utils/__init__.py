from . import synth
