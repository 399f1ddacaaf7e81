"""C# arithmetic in Python: the handful of semantics the codecs depend on."""
import math

INT_MIN = -(1 << 31)


def i32(v):
    """unchecked int arithmetic: wrap to 32 bits, two's complement"""
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v & 0x80000000 else v


def i16(v):
    """(short) of an int: keep the low 16 bits"""
    v &= 0xFFFF
    return v - (1 << 16) if v & 0x8000 else v


def tdiv(a, b):
    """int / int: truncation toward zero"""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def tmod(a, b):
    """int % int: sign follows the dividend"""
    return a - tdiv(a, b) * b


def to_int(x):
    """(int) of a double: truncation; NaN and values outside int32 give 0x80000000 (cvttsd2si)"""
    if x != x or x >= 2147483648.0 or x <= -2147483649.0:
        return INT_MIN
    return int(x)


def clamp(v, lo, hi):
    if v < lo:
        return lo
    if v > hi:
        return hi
    return v


def clamp16(v):
    return clamp(v, -32768, 32767)


def clamp4(v):
    return clamp(v, -8, 7)


def div_round_up(a, b):
    """IntegerExtensions.DivideByRoundUp"""
    return int(math.ceil(a / b)) if b else 0


def next_multiple(value, multiple):
    """Helpers.GetNextMultiple"""
    if multiple <= 0 or value % multiple == 0:
        return value
    return value + multiple - value % multiple


SIGNED_NIBBLE = (0, 1, 2, 3, 4, 5, 6, 7, -8, -7, -6, -5, -4, -3, -2, -1)


def combine_nibbles(high, low):
    return ((high << 4) | (low & 0xF)) & 0xFF
