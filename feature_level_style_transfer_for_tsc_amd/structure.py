"""Omni-scale layer spec (host-side, pure Python) — same public names and results as the reference's
OS_CNN/OS_CNN_Structure_build.py:3-42 and the tap-window helper of OS_CNN/OS_CNN.py:9-12."""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

LayerSpec = List[Tuple[int, int, int]]


def get_Prime_number_in_a_range(start: int, end: int) -> List[int]:
    """Values in [start, end] with no divisor in [2, v) — so 1 is included (Structure_build.py:3-13)."""
    return [v for v in range(start, end + 1) if not any(v % d == 0 for d in range(2, v))]


def get_out_channel_number(paramenter_layer: int, in_channel: int, prime_list: Sequence[int]) -> int:
    return int(paramenter_layer / (in_channel * sum(prime_list)))            # :16-18


def generate_layer_parameter_list(start: int, end: int, paramenter_number_of_layer_list: Sequence[int],
                                  in_channel: int = 1) -> List[LayerSpec]:
    """[[(in, out, kernel) per prime] per budget] + a final two-branch layer (:20-42)."""
    primes = get_Prime_number_in_a_range(start, end)
    if not primes:
        print('start = ', start, 'which is larger than end = ', end)
    layers: List[LayerSpec] = []
    cin = in_channel
    for budget in paramenter_number_of_layer_list:
        width = get_out_channel_number(budget, cin, primes)
        layers.append([(cin, width, p) for p in primes])
        cin = len(primes) * width
    last_width = len(primes) * get_out_channel_number(paramenter_number_of_layer_list[0], in_channel, primes)
    layers.append([(cin, last_width, start), (cin, last_width, start + 1)])
    return layers


def layer_parameter_list_input_change(layer_parameter_list: Sequence[LayerSpec], input_channel: int) -> List[LayerSpec]:
    """Same widths and kernels, first layer re-fed with ``input_channel`` channels (OS_CNN.py:142-152)."""
    first = [(input_channel, out, k) for (_, out, k) in layer_parameter_list[0]]
    return [first] + [list(layer) for layer in layer_parameter_list[1:]]


def calculate_mask_index(kernel_length_now: int, largest_kernel_lenght: int) -> Tuple[int, int]:
    """[lo, hi) of the Kmax window that a ``kernel_length_now``-tap branch occupies (OS_CNN.py:9-12)."""
    right = math.ceil((largest_kernel_lenght - 1) / 2) - math.ceil((kernel_length_now - 1) / 2)
    lo = largest_kernel_lenght - kernel_length_now - right
    return lo, lo + kernel_length_now


def row_live_ranges(layer: LayerSpec) -> List[Tuple[int, int]]:
    """Per output channel live tap range of a packed omni-scale layer (branches concatenated in order)."""
    kmax = layer[-1][2]
    out: List[Tuple[int, int]] = []
    for _, width, k in layer:
        out.extend([calculate_mask_index(k, kmax)] * width)
    return out


def out_channels(layer: LayerSpec) -> int:
    return sum(width for _, width, _ in layer)
