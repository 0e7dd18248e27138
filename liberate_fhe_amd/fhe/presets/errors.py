"""Exception types and the error-logging decorator of the engine API.

Same class names, constructor arguments and message texts as the reference's
src/liberate/fhe/presets/errors.py so that callers catching or printing them behave identically.
"""
import functools
import logging


def log_error(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        try:
            return func(*args, **kwargs)
        except Exception as e:
            logging.error(f"[Error] Error in {func.__name__} : {e}")
            raise

    return wrapper


class _EngineError(Exception):
    """Base: keeps the formatted text in `message_error`, shown by str() and repr()."""

    def __init__(self, message_error: str):
        self.message_error = message_error
        super().__init__(message_error)

    def __repr__(self):
        return repr(self.message_error)

    def __str__(self):
        return self.message_error


class TestException(_EngineError):
    def __init__(self):
        super().__init__("test error")


class KernelLimitExceeded(_EngineError):
    """Not in the reference: a parameter set the fused HIP kernels are not built for (include/ckks_hip.h: lf_limits)."""

    def __init__(self, what: str):
        super().__init__(what)


class NotFoundMessageSpecialPrimes(_EngineError):
    def __init__(self, message_bit, N):
        super().__init__(f"Can't find message_bit = {message_bit:3<d} and N = {N:6<d}".strip())


class NotFoundScalePrimes(_EngineError):
    def __init__(self, scale_bits, N):
        super().__init__(f"Can't find scale bits = {scale_bits:3<d} and N = {N:6<d}".strip())


class NotEnoughPrimes(_EngineError):
    def __init__(self, scale_bits, N):
        super().__init__(f"Not enough scale bit at scale bits = {scale_bits:3<d} and N = {N:6<d}".strip())


class ViolatedAllowedQbits(_EngineError):
    def __init__(self, scale_bits, N, num_scales, max_qbits, total_qbits):
        super().__init__(
            "Maximum allowed qbits are violated:\n"
            f"max_qbits={max_qbits:4d} and the\n"
            f"requested total is {total_qbits:4d}.\n"
            f"scale_bits = {scale_bits:6<d}, N = {N:3<d} and num_scales = {num_scales:4<d}\n"
        )


class NotEnoughPrimesForBiasGuard(_EngineError):
    def __init__(self, bias_guard, num_special_primes):
        super().__init__(
            "Guarding against biased overflow\nrequires the number of special prime\n"
            "channels greater than 2.\n"
            f"bias_guard = {bias_guard}, num_special_primes = {num_special_primes}"
        )


class NotFindBufferBitLength(_EngineError):
    def __init__(self, buffer_bit_length):
        super().__init__(
            f"Can't find buffer length bit {buffer_bit_length}.\nYou can only choose between 30 or 62."
        )


class SecretKeyNotIncludeSpecialPrime(_EngineError):
    def __init__(self):
        super().__init__("The input secret key must include special prime channels.")


class DifferentTypeError(_EngineError):
    def __init__(self, a, b):
        super().__init__(f"The data type are different. {a}, {b}")


class NotMatchType(_EngineError):
    def __init__(self, origin, to):
        super().__init__(f"The data_struct origin should be a '{to}', but it is '{origin}'.")


class NotMatchDataStructState(_EngineError):
    def __init__(self, origin: str):
        super().__init__(
            f"Wrong format of the source {origin} detected, \n"
            "Apply ntt and the montgomery transformation to the data."
        )


class MaximumLevelError(_EngineError):
    def __init__(self, level, level_max):
        super().__init__(
            "The number of multiplications available\n"
            "for this cipher text is fully depleted. \n"
            "I cannot proceed further.\n"
            f"maximum : {level_max:2d}, now : {level:2d}".strip()
        )


class DeviceSelectError(_EngineError):
    def __init__(self):
        super().__init__("To download data to the CPU, it must already be in a GPU!!!")
