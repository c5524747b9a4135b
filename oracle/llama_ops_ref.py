"""CPU oracle for RoPE and RMSNorm -- TEST INFRASTRUCTURE, NOT PRODUCT.

Restates lwm/llama.py:320-375 in numpy.  PINNED (round 5) to a run of the reference's own lines:
tests/golden/gen_ref_run_golden.py executes precompute_freqs_cis, apply_rotary_emb and RMSNorm._norm / __call__ where they lie
in /root/reference (numpy standing in for the jax.numpy names they use) and commits tests/golden/ref_run.npz;
tests/test_golden.py holds precompute_freqs_cis / apply_rotary_emb below to those vectors bit for bit and rmsnorm to 4 f32
ulps (the mean of squares is taken in float64 here).  XLA's own rounding is not pinned (jax cannot be imported here).  The
cross-entropy below restates tux (absent): unpinned.  float32 where the reference computes in float32; float64 variants
are provided for gradient checks.
"""
import numpy as np

from .attention_ref import round_bf16


def precompute_freqs_cis(dim, max_pos, theta=10000.0, dtype=np.float32):
    """lwm/llama.py:344-350, returning complex64."""
    freqs = 1.0 / (theta ** (np.arange(0, dim, 2)[: (dim // 2)].astype(dtype) / dim))
    t = np.arange(max_pos)
    freqs = np.outer(t, freqs).astype(dtype)
    return np.complex64(np.cos(freqs) + 1j * np.sin(freqs))


def apply_rotary_emb(x, freqs_cis, position_ids, out_bf16=True):
    """lwm/llama.py:353-375 for one tensor: x (B,S,H,D) -> f32 complex multiply on
    interleaved pairs, result cast to the model dtype (bf16 when out_bf16)."""
    x = np.asarray(x, np.float32)
    fc = freqs_cis[np.asarray(position_ids)]            # jnp.take, lwm/llama.py:515
    xr = x.reshape(x.shape[:-1] + (-1, 2))
    xc = xr[..., 0] + 1j * xr[..., 1]
    out = xc.astype(np.complex64) * fc[:, :, None, :]
    y = np.stack((out.real, out.imag), axis=-1).reshape(x.shape).astype(np.float32)
    return round_bf16(y) if out_bf16 else y


def rope_bwd(g, freqs_cis, position_ids):
    """Gradient of apply_rotary_emb w.r.t. x: multiply by the conjugate."""
    return apply_rotary_emb(g, np.conj(freqs_cis), position_ids, out_bf16=False)


def rmsnorm(x, weight, eps=1e-6, out_bf16=True):
    """lwm/llama.py:335-341: f32 upcast, x*rsqrt(mean(x^2)+eps) cast to dtype, times weight (dtype)."""
    x32 = np.asarray(x, np.float32)
    r = 1.0 / np.sqrt(np.mean(np.square(x32.astype(np.float64)), axis=-1, keepdims=True) + eps)
    y = (x32 * r).astype(np.float32)
    if not out_bf16:
        return y * np.asarray(weight, np.float32)
    return round_bf16(round_bf16(y) * round_bf16(np.asarray(weight, np.float32)))


def rmsnorm_bwd(x, weight, g, eps=1e-6):
    """float64 gradients of out = (x * rsqrt(mean(x^2)+eps)) * w (casts treated as identity)."""
    x, w, g = (np.asarray(a, np.float64) for a in (x, weight, g))
    Cc = x.shape[-1]
    r = 1.0 / np.sqrt(np.mean(x * x, axis=-1, keepdims=True) + eps)
    xh = x * r
    dy = g * w
    dx = r * (dy - xh * np.sum(dy * xh, axis=-1, keepdims=True) / Cc)
    dw = np.sum((g * xh).reshape(-1, Cc), axis=0)
    return dx, dw


def cross_entropy_loss_and_accuracy(logits, tokens, valid=None):
    """tux.cross_entropy_loss_and_accuracy [upstream package, restated from its published source;
    call sites lwm/train.py:177-181, :192-201], float64.  Returns (loss, accuracy, dloss/dlogits)."""
    x = np.asarray(logits, np.float64)
    B, S, V = x.shape
    tokens = np.asarray(tokens)
    valid = np.ones((B, S)) if valid is None else np.asarray(valid, np.float64)
    length = np.maximum(valid.sum(axis=-1), 1e-10)
    m = x.max(axis=-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(x - m).sum(axis=-1))
    logp_t = np.take_along_axis(x, tokens[..., None], axis=-1)[..., 0] - lse
    logp_t = np.where(valid > 0, logp_t, 0.0)
    loss = -np.mean(np.sum(logp_t, axis=-1) / length)
    correct = np.where(valid > 0, x.argmax(axis=-1) == tokens, False)
    acc = np.mean(np.sum(correct, axis=-1) / length)
    p = np.exp(x - lse[..., None])
    onehot = np.zeros_like(p)
    np.put_along_axis(onehot, tokens[..., None], 1.0, axis=-1)
    w = np.where(valid > 0, 1.0, 0.0) / (length[:, None] * B)
    return loss, acc, (p - onehot) * w[..., None]


def swiglu(a, b):
    """silu(a) * b in float64 (lwm/llama.py:659) and its gradients given g."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return a / (1.0 + np.exp(-a)) * b


def swiglu_bwd(a, b, g):
    a, b, g = (np.asarray(t, np.float64) for t in (a, b, g))
    sg = 1.0 / (1.0 + np.exp(-a))
    return g * b * sg * (1.0 + a * (1.0 - sg)), g * a * sg
