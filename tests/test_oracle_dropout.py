"""CPU checks of the train-mode mask restatement (oracle/pgrm.py drop_mask, the numpy twin of csrc/common.h drop_scale):
known-answer values of the splitmix64 finaliser, keep rate, independence across seeds, and that the oracle's training
forward reduces to the pinned eval forward at p = 0."""
import numpy as np
import torch

from dpmn_amd.utils import synth
from helpers import load_golden, sd_from_manifest


def _splitmix_ref(seed, idx):
    """python-int restatement (arbitrary precision, masked to 64 bits) -- independent of numpy's wraparound"""
    M = (1 << 64) - 1
    z = (idx * 0x9E3779B97F4A7C15 + seed) & M
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & M
    z ^= z >> 31
    return z


def test_drop_mask_known_answers_and_rate():
    from oracle import pgrm as o
    # splitmix64's published first output for state 0 (idx = 1, seed = 0 is one state increment): 0xE220A8397B1DCDAF
    assert _splitmix_ref(0, 1) == 0xE220A8397B1DCDAF
    seed, p = 0x1234567890ABCDE, 0.1
    idx = np.array([0, 1, 2, 3, 1000, 2 ** 33 + 5, 2 ** 40 + 7], dtype=np.uint64)
    want = [(1.0 / (1.0 - np.float32(p))) if np.float32((_splitmix_ref(seed, int(i)) >> 40) * 2.0 ** -24) >= np.float32(p) else 0.0
            for i in idx]
    got = o.drop_mask(seed, idx, p)
    assert np.allclose(got.numpy(), np.array(want, dtype=np.float32), rtol=0, atol=0)
    m = o.drop_mask(seed, np.arange(1 << 20), p)
    assert abs(float((m != 0).float().mean()) - 0.9) < 2e-3
    m2 = o.drop_mask(seed + 1, np.arange(1 << 20), p)
    assert abs(float(((m != 0) == (m2 != 0)).float().mean()) - 0.82) < 5e-3      # independent: 0.9^2 + 0.1^2


def test_oracle_training_forward_reduces_to_eval_at_zero_rates():
    from oracle import pgrm as o
    g = load_golden("pgrm_mode1_iter2")
    B, _, _, wseed, iseed = [int(v) for v in g["meta"]]
    sd = sd_from_manifest(g["manifest"], wseed)
    x_q = (synth.uniform("x_q", (B, 1, 32, 128), 0, 1, iseed) > 0.5).float().repeat(1, 3, 1, 1)
    x_kv = synth.uniform("x_kv", (B, 3, 32, 128), 0, 1, iseed)
    res = [synth.uniform("res%d" % i, (B, 3, 32, 128), 0, 1, iseed) for i in range(2)]
    ev = o.pgrm_forward(sd, x_q, x_kv, res)
    drop = dict(p=0.0, pa=0.0, dp=(0.0, 0.0), seeds=list(range(12)))
    assert torch.equal(o.pgrm_forward(sd, x_q, x_kv, res, drop=drop), ev)
    drop = dict(p=0.1, pa=0.1, dp=(0.05, 0.06), seeds=list(range(100, 112)))
    out = o.pgrm_forward(sd, x_q, x_kv, res, drop=drop)
    assert torch.isfinite(out).all() and not torch.equal(out, ev)
