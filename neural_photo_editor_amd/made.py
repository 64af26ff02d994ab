"""MADE mask generation on the host (product code; bit-exact requirement of the north star).

Restates mask_generator.py:15-103 for the only configuration the reference uses
(MADE(hidden_sizes=[n]), mask_distribution=0, random_seed=1234: layers.py:756-759, IAN.py:127):

  * reset("Once") (layers.py:845-853, called from API.py:33-36 and train_IAN.py:404) resets the
    ordering to arange, re-seeds the RandomStreams (mask_generator.py:56-60) and performs exactly one
    shuffle_ordering() + sample_connectivity();
  * shuffle_ordering() = RandomStreams(seed).shuffle_row_elements(ordering) (:35): a numpy
    RandomState permutation whose seed is RandomState(seed).randint(2**30) [recalled Theano
    shared_randomstreams semantics]; numpy's legacy RandomState stream is frozen, so this is exact;
  * with l = 0 the multinomial of _get_hidden_layer_connectivity (:81-91) is one-hot, so every hidden
    unit gets connectivity min(ordering + 1) = 1 -- no MRG31k3p draw influences the masks
    (SURVEY App. C);
  * _get_mask (:93-94): M[i, j] = (c_in[i] <= c_out[j]).

Both MADEs of the IAN (l_IAF_mu, l_IAF_ls) use the default seed and therefore share these masks.
"""
from __future__ import annotations

import numpy as np

DEFAULT_SEED = 1234  # mask_generator.py:17


def shuffled_ordering(input_size, random_seed=DEFAULT_SEED):
    child_seed = int(np.random.RandomState(random_seed).randint(2 ** 30))
    perm = np.random.RandomState(child_seed).permutation(input_size)
    return np.arange(input_size, dtype=np.int64)[perm]


def masks_once(input_size, hidden_size=None, random_seed=DEFAULT_SEED):
    """(M_input, M_output, M_direct) as float32 0/1 arrays laid out (in, out) like the weights."""
    hidden_size = input_size if hidden_size is None else hidden_size
    ordering = shuffled_ordering(input_size, random_seed)
    conn_in = ordering + 1                                   # layers_connectivity[0]  (:36)
    conn_hidden = np.full((hidden_size,), conn_in.min(), np.int64)  # one-hot multinomial, l = 0
    conn_out = ordering                                      # layers_connectivity[-1] (:31)
    m_input = (conn_in[:, None] <= conn_hidden[None, :]).astype(np.float32)
    m_output = (conn_hidden[:, None] <= conn_out[None, :]).astype(np.float32)
    m_direct = (conn_in[:, None] <= conn_out[None, :]).astype(np.float32)
    return m_input, m_output, m_direct
