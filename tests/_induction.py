"""Hand-set weights for the 2-layer harness that implement an induction (copy) circuit -- the
synthetic stand-in for the reference's needle evaluation (scripts/eval_needle.py needs trained
weights).

  layer 1, head 0  "previous token": q and k are constants living on the FASTEST RoPE pairs,
                   k pre-rotated by one step, so that after RoPE q_i . k_j = g1^2 * sum_p
                   cos(w_p (j - i + 1)): sharply peaked at j = i - 1.  The value path copies the
                   token-identity code of position j into the PREV sub-space of the residual.
  layer 2, head 0  "induction": q = identity code of the current token, k = PREV code, both on
                   the SLOWEST RoPE pairs (w_p * distance << 1 -- which is exactly why the
                   reference raises theta to 1e7 / 5e7 for 128K / 1M contexts, README.md:112-117),
                   so position j scores high iff token[j-1] == token[i].  The value path copies
                   the identity code of token[j] into the OUT sub-space, which lm_head decodes.
  head 1 of both layers and the MLP are zero.

Residual layout (hidden 256): ID 0:64 (+-1 Hadamard code of the token), PREV 64:128, OUT 128:192,
CONST 192 (=8, the handle the positional head projects from).  With the needle pair (K, V)
planted once at depth d and K as the final token, the argmax at the last position must be V.
"""
import math

import numpy as np
import torch

HID, H, D, VOCAB = 256, 2, 128, 32
ID, PREV, OUT, CONST = 0, 64, 128, 192
N_FAST, N_SLOW = 8, 13          # RoPE pairs used by the positional / the content match
KEY_TOKEN, VALUE_TOKEN = 1, 2


def codes(seed=0):
    """Rows 0..31 of the 64x64 Sylvester-Hadamard matrix: orthogonal over the 64 ID dims (lm_head
    reads 64 on a hit, 0 otherwise) and, restricted to the first 26 dims that the content match
    uses, rows of H32 cut to 26 columns: 26 on a hit, |.| <= 6 otherwise."""
    r = np.arange(64)
    h = np.array([[(-1.0) ** bin(a & b).count("1") for b in r] for a in r], np.float32)
    c = h[:VOCAB]
    cross = c[:, :2 * N_SLOW] @ c[:, :2 * N_SLOW].T
    np.fill_diagonal(cross, -99)
    assert cross.max() <= 6
    return c


def _freqs(theta):
    return theta ** (-np.arange(0, D, 2, dtype=np.float64) / D)


def positional_margin(theta, S):
    """(score at j = i-1) - (best score elsewhere) of the previous-token head, in units of g1^2."""
    w = _freqs(theta)[:N_FAST]
    delta = np.arange(-S, 1, dtype=np.float64)[:, None] + 1.0          # j - i + 1 for j <= i
    s = np.cos(delta * w[None]).sum(-1)
    best = s[-2]                                                       # j = i - 1
    s[-2] = -np.inf
    return best - s.max()


def build(theta, S, seed=0):
    """-> (LLaMAConfig kwargs, {harness parameter name: float32 tensor})."""
    c = codes(seed)
    w = _freqs(theta)
    m1 = positional_margin(theta, S)
    assert m1 > 0.05, m1
    # logit gap wanted: ln(S) + 12 (a 1e-5 share for everything else at most)
    gap = math.log(S) + 12.0
    g1 = math.sqrt(gap * math.sqrt(D) / m1)
    hit = sum(2 * math.cos(w[D // 2 - N_SLOW + p] * S) for p in range(N_SLOW))     # worst-case rotation at distance S
    assert hit > 16, hit
    g2 = math.sqrt(gap * math.sqrt(D) / (hit - 6))
    st = {}
    emb = np.zeros((VOCAB, HID), np.float32)
    emb[:, ID:ID + 64] = c
    emb[:, CONST] = 8.0
    st["wte"] = emb
    rms1 = math.sqrt((64 + 64.0) / HID)                      # |code|^2 + CONST^2 over 256
    rms2 = math.sqrt((64 + 64 + 64.0) / HID)                 # + PREV
    z = lambda *s: np.zeros(s, np.float32)
    for i in range(2):
        p = f"h.{i}."
        for n in ("wq", "wk", "wv", "wo"):
            st[p + "attention." + n] = z(HID, HID)
        st[p + "feed_forward.w1"], st[p + "feed_forward.w3"] = z(HID, 512), z(HID, 512)
        st[p + "feed_forward.w2"] = z(512, HID)
        st[p + "attention_norm.kernel"], st[p + "ffn_norm.kernel"] = np.ones(HID, np.float32), np.ones(HID, np.float32)
    st["ln_f.kernel"] = np.ones(HID, np.float32)
    # ---- layer 1, head 0: previous-token head.  normalised CONST channel = 8 / rms1
    cn = 8.0 / rms1
    for p in range(N_FAST):
        st["h.0.attention.wq"][CONST, 2 * p] = g1 / cn                       # q pair = g1 (1, 0)
        st["h.0.attention.wk"][CONST, 2 * p] = g1 * math.cos(w[p]) / cn      # k pair = g1 R(w_p)(1, 0)
        st["h.0.attention.wk"][CONST, 2 * p + 1] = g1 * math.sin(w[p]) / cn
    for k in range(64):
        st["h.0.attention.wv"][ID + k, k] = 1.0                              # v = code / rms1
        st["h.0.attention.wo"][k, PREV + k] = rms1                           # PREV <- code of the attended token
    # ---- layer 2, head 0: induction head on the slowest pairs (head dims 2*(64-N_SLOW) .. 127)
    base = 2 * (D // 2 - N_SLOW)
    for m in range(2 * N_SLOW):
        st["h.1.attention.wq"][ID + m, base + m] = g2 * rms2                 # q = g2 * code(token_i)
        st["h.1.attention.wk"][PREV + m, base + m] = g2 * rms2               # k = g2 * code(token_{j-1})
    for k in range(64):
        st["h.1.attention.wv"][ID + k, k] = 1.0
        st["h.1.attention.wo"][k, OUT + k] = rms2
    head = z(HID, VOCAB)
    head[OUT:OUT + 64] = c.T                                                 # logit[t] = OUT . code(t)
    st["lm_head"] = head
    cfg = dict(vocab_size=VOCAB, hidden_size=HID, intermediate_size=512, num_hidden_layers=2,
               num_attention_heads=H, max_sequence_length=S, theta=theta, rms_norm_eps=1e-6)
    return cfg, {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items()}


def haystack(S, depth, seed=1):
    """BOS (token 0; position 0 has no predecessor, so its PREV slot holds its own code), random
    tokens from {3..VOCAB-1}, (KEY, VALUE) planted at `depth`, KEY again as the last token."""
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(3, VOCAB, (1, S), generator=g)
    t[0, 0] = 0
    pos = 1 + min(S - 4, max(0, int(depth * (S - 4))))
    t[0, pos], t[0, pos + 1] = KEY_TOKEN, VALUE_TOKEN
    t[0, S - 1] = KEY_TOKEN
    return t, pos
