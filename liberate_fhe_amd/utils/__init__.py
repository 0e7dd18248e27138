from . import helpers, synth
