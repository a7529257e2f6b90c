import math


def nearest_power_of_two_half(x):
    """2 ** round(log2(x / 2)), at least 1 ('auto' embedding width)."""
    return int(math.ceil(2 ** max(round(math.log(0.5 * x, 2)), 0)))


def nearest_power_of_two(x):
    """Smallest power of two >= x ('auto' input-mapping width)."""
    return int(math.ceil(2 ** max(int(math.ceil(math.log(x, 2))), 0)))
