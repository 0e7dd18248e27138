"""Named parameter presets (reference: src/liberate/fhe/presets/params.py:1-30)."""


def _preset(logN, num_special_primes, devices):
    return {
        "logN": logN,
        "num_special_primes": num_special_primes,
        "devices": devices,
        "scale_bits": 40,
        "num_scales": None,
    }


params = {
    "bronze": _preset(14, 1, [0]),
    "silver": _preset(15, 2, [0]),
    "gold": _preset(16, 4, None),
    "platinum": _preset(17, 6, None),
}
