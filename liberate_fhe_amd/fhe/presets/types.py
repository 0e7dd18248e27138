"""Origin tags carried by every data_struct (reference: src/liberate/fhe/presets/types.py:1-11)."""

origins = dict(
    sk="secret key",
    pk="public key",
    ksk="key switch key",
    rotk="rotation key:",
    galk="galois key",
    conjk="conjugation key",
    ct="cipher text",
    ctt="cipher text triplet",
)
