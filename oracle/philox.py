"""The draw table of the library's in-kernel random numbers (include/idealnerf.h: idealnerf_philox_uniform,
idn_render_args.rng_mode), restated on the CPU with numpy integer arithmetic.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.

The reference has no counterpart to pin this against: it draws ``t_rand = torch.rand(z_vals.shape)``
(NeRFs/HeadNeRF/train/audio_exp_nerf.py:314-326) and ``u = torch.rand(list(cdf.shape[:-1]) + [N_samples])``
(NeRFs/HeadNeRF/helper.py:283) from torch's generator, whose numbers on a GPU depend on its launch geometry.  What is
pinned here is the generator itself: Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
1, 2, 3", SC'11), against the three known-answer vectors its authors publish (Random123 ``kat_vectors``, philox4x32 10
rounds) -- ``tests/test_oracle_golden.py::test_philox_known_answers``.  The renderer's parity with these draws is then
the usual one: the oracle's render_rays fed this table as ``t_rand`` / ``u``.
"""
import numpy as np

__all__ = ["philox4x32_10", "uniform_table", "KNOWN_ANSWERS"]

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)

# (counter, key) -> output, Random123 kat_vectors "philox4x32 10"
KNOWN_ANSWERS = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(counter, key):
    """counter: four arrays (or ints) of 32-bit words, key: two 32-bit words -> four uint64 arrays of 32-bit words."""
    c = [np.asarray(x, dtype=np.uint64) & _LO for x in counter]
    k0, k1 = np.uint64(key[0]) & _LO, np.uint64(key[1]) & _LO
    for _ in range(10):
        p0, p1 = _M0 * c[0], _M1 * c[2]          # 32 x 32 -> 64 bit products (no overflow in uint64)
        c = [(p1 >> _S32) ^ c[1] ^ k0, p1 & _LO, (p0 >> _S32) ^ c[3] ^ k1, p0 & _LO]
        k0, k1 = (k0 + _W0) & _LO, (k1 + _W1) & _LO
    return c


def uniform_table(seed: int, which: int, row0: int, n_rows: int, n_cols: int) -> np.ndarray:
    """out[r, c] = 24-bit uniform of word c % 4 of Philox4x32-10(key = seed, counter = (c // 4, 0, 2 (row0 + r) + which)):
    which = 0 the stratified offsets (t_rand), 1 the importance draws (u).  float32 [n_rows, n_cols]."""
    rows = (np.arange(n_rows, dtype=np.uint64) + np.uint64(row0)) * np.uint64(2) + np.uint64(which)
    blocks = np.arange((n_cols + 3) // 4, dtype=np.uint64)
    r, b = np.meshgrid(rows, blocks, indexing="ij")
    words = philox4x32_10((b, np.zeros_like(b), r & _LO, r >> _S32), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    x = np.stack(words, axis=-1).reshape(n_rows, -1)[:, :n_cols]          # word w of block b is column 4 b + w
    return ((x >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)
