"""Dense-mask / split-K forward (ringattention_inference flavour), the partial
combine and the KV-cache write, emulated on the host through the C ABI."""
import numpy as np
import pytest

from oracle import attention_ref as R
from tests import _emu


def _rnd(shape, seed):
    return R.round_bf16(np.random.default_rng(seed).standard_normal(shape).astype(np.float32))


@pytest.mark.parametrize("B,Q,K,H,splits,cache_index", [
    (1, 1, 300, 2, 1, 200),      # decode, one piece
    (2, 1, 520, 1, 3, 519),      # decode, split-K with a ragged last piece
    (1, 5, 257, 1, 2, 100),      # short block of queries (q_len != kv_len)
    (1, 1, 130, 1, 4, 10),       # pieces that are entirely masked (cache mostly empty)
])
def test_decode_mask_splitk(B, Q, K, H, splits, cache_index):
    q, k, v = _rnd((B, Q, H, 128), 1), _rnd((B, K, H, 128), 2), _rnd((B, K, H, 128), 3)
    am = (np.random.default_rng(4).random((B, K)) > 0.1).astype(np.uint8)
    am[:, cache_index] = 1
    mask = R.decode_mask(B, Q, K, cache_index, am)
    out, lse = _emu.attn_infer(q, k, v, mask, k_splits=splits)
    ro, rl = R.dense_attention(q, k, v, causal=False, dense_mask=mask)
    assert np.abs(out - ro).max() / np.abs(ro).max() < 1e-2
    assert np.abs(lse - rl).max() < 1e-4
    # the f32 ring restatement of ringattention_inference agrees as well
    assert np.abs(R.ring_inference(q, k, v, mask, ring=1) - ro).max() < 1e-5


def test_decode_streams_only_the_visible_range():
    """Left padding + a mostly empty cache: the decode kernel scans its mask piece and reads K/V only in
    [first visible, last visible]; pieces before / after / inside the range must still combine exactly."""
    B, Q, K, H = 2, 1, 1000, 2
    q, k, v = _rnd((B, Q, H, 128), 21), _rnd((B, K, H, 128), 22), _rnd((B, K, H, 128), 23)
    am = np.ones((B, K), np.uint8)
    am[0, :130] = 0                      # left-padded prompt in row 0
    am[1, 37] = 0                        # a hole in row 1
    mask = R.decode_mask(B, Q, K, 333, am)
    for splits in (1, 7, 16):
        out, lse = _emu.attn_infer(q, k, v, mask, k_splits=splits)
        ro, rl = R.dense_attention(q, k, v, causal=False, dense_mask=mask)
        assert np.abs(out - ro).max() / np.abs(ro).max() < 1e-2, splits
        assert np.abs(lse - rl).max() < 1e-4, splits
    # poison everything outside the visible range: the result must not change (nothing is read there)
    k2, v2 = k.copy(), v.copy()
    k2[0, :130], v2[0, :130], k2[:, 334:], v2[:, 334:] = 1e30, 1e30, 1e30, 1e30
    out2, _ = _emu.attn_infer(q, R.round_bf16(k2), R.round_bf16(v2), mask, k_splits=7)
    out1, _ = _emu.attn_infer(q, k, v, mask, k_splits=7)
    assert np.array_equal(out1, out2)


def test_fully_masked_rows_and_arbitrary_mask():
    B, Q, K, H = 1, 4, 128, 1
    q, k, v = _rnd((B, Q, H, 128), 5), _rnd((B, K, H, 128), 6), _rnd((B, K, H, 128), 7)
    mask = (np.random.default_rng(8).random((B, Q, K)) > 0.5).astype(np.uint8)
    mask[:, 2] = 0                       # a query that sees nothing
    out, lse = _emu.attn_infer(q, k, v, mask, k_splits=2)
    ro, rl = R.dense_attention(q, k, v, causal=False, dense_mask=mask)
    assert np.all(out[:, 2] == 0) and np.isneginf(lse[0, 0, 2])
    assert np.abs(out - ro).max() / np.abs(ro).max() < 1e-2
    fin = np.isfinite(rl)
    assert np.abs(lse[fin] - rl[fin]).max() < 1e-4


def test_kv_cache_write():
    B, S, H, D = 2, 40, 2, 128
    cache = _emu.bf16_array(np.zeros((B, S, H, D), np.float32))
    new = _emu.bf16_array(_rnd((B, 7, H, D), 9))
    _emu.kv_cache_write(cache, new, dst_row0=30, src_row0=2, nrows=5)
    assert np.array_equal(cache[:, 30:35], new[:, 2:7])
    assert not cache[:, :30].any() and not cache[:, 35:].any()


def test_kv_cache_write_at_device_index():
    """lwm_kv_cache_write_at: the row comes from (device) memory; rows outside the shard are skipped --
    the decode rule of lwm/llama.py:454-467 (shard r of a cache sharded in blocks of S rows)."""
    B, S, H, D = 2, 40, 2, 128
    new = _emu.bf16_array(_rnd((B, 3, H, D), 11))
    cache = _emu.bf16_array(np.zeros((B, S, H, D), np.float32))
    _emu.kv_cache_write_at(cache, new, index=17, row_offset=0, src_row0=0, nrows=3)
    assert np.array_equal(cache[:, 17:20], new) and not cache[:, :17].any() and not cache[:, 20:].any()
    # global row 41 on shard 1 (rows 40..79): local row 1; on shard 0 nothing is written
    c0 = _emu.bf16_array(np.zeros((B, S, H, D), np.float32))
    c1 = _emu.bf16_array(np.zeros((B, S, H, D), np.float32))
    _emu.kv_cache_write_at(c0, new, index=41, row_offset=0, src_row0=0, nrows=1)
    _emu.kv_cache_write_at(c1, new, index=41, row_offset=-S, src_row0=0, nrows=1)
    assert not c0.any() and np.array_equal(c1[:, 1:2], new[:, :1]) and not c1[:, 2:].any()
    # a 3-row write that straddles the end of the shard keeps only the rows inside
    c2 = _emu.bf16_array(np.zeros((B, S, H, D), np.float32))
    _emu.kv_cache_write_at(c2, new, index=S - 2, row_offset=0, src_row0=0, nrows=3)
    assert np.array_equal(c2[:, S - 2:], new[:, :2])
