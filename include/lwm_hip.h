/* lwm_hip.h -- C ABI of the MI355X-native LWM hot path (liblwm_hip.so).
 *
 * The reference (LargeWorldModel/LWM) has no FFI: its hot path sits behind
 * Python callables.  Each entry point below names the reference interface it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*; NULL = default stream);
 *   - every pointer is DEVICE memory owned by the caller; the library never
 *     allocates, frees or synchronises; kernels are enqueued on `stream`;
 *   - return 0 (LWM_OK) or a negative LWM_E* code; lwm_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - no global mutable state; one host thread per device is the supported
 *     threading model.
 */
#ifndef LWM_HIP_H
#define LWM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LWM_OK 0
#define LWM_EINVAL (-1)       /* bad argument (null, misaligned, bad shape) */
#define LWM_EUNSUPPORTED (-2) /* valid but not implemented (e.g. head_dim != 128) */
#define LWM_ELAUNCH (-3)      /* HIP reported a launch error */

/* A [B, S, H, D] tensor with D contiguous; strides in ELEMENTS.  This is the
 * reference's q/k/v layout: heads are split by reshape, not transposed
 * (lwm/llama.py:434-438), and shard_map hands the op (B, S/sp, H/tp, D) blocks
 * (lwm/llama.py:559-566). */
typedef struct LwmTensor4 {
    void* ptr;
    int64_t stride_b, stride_s, stride_h;
} LwmTensor4;

/* One ring step of blockwise attention: local queries [q_start, q_start+Sq)
 * against the K/V block holding global positions [k_start, k_start+Sk).
 *
 * Replaces: ringattention(q,k,v,attn_bias,segment_ids, axis_name="sp",
 *   float32_logits=True, blockwise_kwargs=dict(causal_block_size=1, ...))
 *   -- call site lwm/llama.py:539-569 -- and its custom-VJP backward.
 * Mask semantics (lwm/llama.py:425, :527-537, :546, :572-592):
 *   visible(q,k) = (k_pos <= q_pos if causal) AND segment_ids_q[q]==segment_ids_k[k]
 *                  AND key_valid[k] != 0
 *   (attn_bias in the reference is the key-padding mask turned into
 *   {0, finfo.min}; key_valid is that mask before the transform).
 * Rows with no visible key produce out = 0, lse = -inf.
 * Scores are q.k * scale in f32; softmax and all accumulators are f32
 * (float32_logits=True, lwm/llama.py:543); operands are bf16; D must be 128.
 */
#define LWM_MAX_PIECES 8
typedef struct LwmAttnArgs {
    LwmTensor4 q, k, v;    /* bf16 in */
    LwmTensor4 out;        /* bf16: fwd writes it when final_out; bwd reads it */
    float* lse;            /* [B,H,Sq] natural-log LSE: fwd writes when final_out; bwd reads */
    float* out_acc;        /* [B,Sq,H,D] f32 dense ring carry (normalised partial out) */
    float* lse_acc;        /* [B,H,Sq]   f32 ring carry */
    LwmTensor4 dout;       /* bf16 in (bwd) */
    LwmTensor4 dq, dk, dv; /* bf16 out (bwd, when final_out) */
    float* delta;          /* backward row statistics, lwm_attn_bwd_delta_bytes(B,H,Sq) bytes, written by
                            * lwm_attn_bwd_delta from out, dout and lse, read by lwm_attn_bwd_dq / _dkdv: per (b,h)
                            * [-lse*log2(e) | -rowsum(dout*out)], each row padded to a multiple of 64 queries */
    float* dq_acc;         /* [B,Sq,H,D] f32 dense carry */
    float* dk_acc;         /* [B,Sk,H,D] f32 dense carry (travels with the K/V block) */
    float* dv_acc;
    const int32_t* segment_ids_q; /* [B,Sq] or NULL */
    const int32_t* segment_ids_k; /* [B,Sk] or NULL (both or neither) */
    const uint8_t* key_valid;     /* [B,Sk] or NULL */
    int32_t B, H, Sq, Sk, D;
    int64_t q_start, k_start; /* global token position of local row 0 */
    float scale;              /* 1/sqrt(D) in the reference */
    int32_t causal;
    int32_t carry_in;  /* 1: merge with the *_acc carries before writing */
    int32_t final_out; /* 1: write bf16 out/lse (fwd) or dq / dk,dv (bwd); 0: write *_acc */
    /* Forward only -- the dense-mask / decode flavour (ringattention_inference,
     * lwm/llama.py:571-614): an arbitrary boolean mask (B,1,Q,K) as u8, combined
     * (AND) with the masks above; element (b,q,k) of THIS K/V block is
     * dense_mask[b*mask_stride_b + q*mask_stride_q + k].  NULL = none. */
    const uint8_t* dense_mask;
    int64_t mask_stride_b, mask_stride_q;
    /* Forward only -- split-K ("flash decoding") for short query blocks: the key
     * range is cut into k_splits contiguous pieces, each handled by its own
     * workgroups; piece s writes its normalised partial to
     * out_acc + s*B*Sq*H*D and lse_acc + s*B*H*Sq (requires final_out = 0,
     * carry_in = 0); merge with lwm_attn_combine.  0 or 1 = no split. */
    int32_t k_splits;
    /* Optional: block-sparsity hints for packed sequences.  seg_blocks_q [B][ceil(Sq/32)][2]
     * and seg_blocks_k [B][ceil(Sk/32)][2] hold the (min, max) segment id of every 32-row
     * block (filled by lwm_attn_segment_blocks; padded keys excluded).  When both are given
     * a workgroup only walks the tiles of the other operand whose segment range meets its
     * own -- whole documents of a packed batch are skipped instead of computed and masked.
     * Results are unchanged (the per-element mask is still applied).  NULL = no skipping. */
    const int32_t* seg_blocks_q;
    const int32_t* seg_blocks_k;
    /* lwm_attn_bwd_dq: 0 = dq_acc is [B,Sq,H,D], 1 = head-major [B,H,Sq,D]. */
    int32_t dq_acc_head_major;
    /* Piecewise position maps (lwm_version() >= 500; *_pieces = 0 or 1: one piece, the fields above say everything).
     * A shard under zigzag ownership is TWO runs of consecutive positions (half-chunks r and 2n-1-r), under a balanced
     * ownership of packed documents a few more, and so is the K/V a rank has gathered from its peers (what lies below its
     * first run, between its runs ...).  With q_pieces = P > 1:
     *   query rows [q_piece_row[i], q_piece_row[i+1]) sit at positions q_piece_pos[i] + (row - q_piece_row[i]),
     *   q_piece_row[0] = 0, q_piece_pos[0] = q_start, the last piece ends at Sq;  keys likewise.
     * One launch then covers what took one launch per (q segment, k segment) pair, with no carry in between -- at small
     * shards (S = 32768 over 8 ranks: 2048-row half-chunks) the pair launches cannot fill 256 CUs.  Requirements:
     * P <= LWM_MAX_PIECES; piece rows are multiples of 256, strictly ascending, inside (0, S); positions ascend with the
     * row and pieces do not overlap (pos[i+1] >= pos[i] + row[i+1] - row[i]).  Training kernels only (lwm_attn_fwd
     * without dense_mask / k_splits, lwm_attn_bwd_dq / _dkdv); results are bitwise those of the single-piece launch on the
     * same rows when the pieces happen to be adjacent. */
    int32_t q_pieces, k_pieces;
    int32_t q_piece_row[LWM_MAX_PIECES], k_piece_row[LWM_MAX_PIECES];
    int64_t q_piece_pos[LWM_MAX_PIECES], k_piece_pos[LWM_MAX_PIECES];
    /* Size in bytes of the buffer behind `delta`, or 0 = not given.  When given, lwm_attn_bwd_delta / _dq / _dkdv
     * refuse a buffer smaller than lwm_attn_bwd_delta_bytes(B,H,Sq) instead of overrunning it (the layout of the
     * statistics changed at lwm_version() 400: a [B,H,Sq] buffer of earlier versions is too small). */
    int64_t delta_bytes;
} LwmAttnArgs;

int lwm_attn_fwd(const LwmAttnArgs* args, void* stream);
/* The backward of one ring step = three launches (no atomics, bit-reproducible): the row statistics once per query
 * block, then dq (a workgroup owns 128 queries and streams K/V: 3 GEMM units) and dk, dv (a workgroup owns 128 keys
 * and streams Q/dO: 4 units) per (query block, K/V block) pair, chained through the f32 carries.  [A form that
 * computed S and dP once and summed bf16 dq partials -- 5 units -- was measured in rounds 2-3 and retired in round 4:
 * profiles/r04_backward.md.] */
int lwm_attn_bwd_delta(const LwmAttnArgs* args, void* stream);
int lwm_attn_bwd_dq(const LwmAttnArgs* args, void* stream);
int lwm_attn_bwd_dkdv(const LwmAttnArgs* args, void* stream);

/* Bytes of the backward's row statistics (LwmAttnArgs::delta) for a [B, Sq, H, D] query block. */
int64_t lwm_attn_bwd_delta_bytes(int32_t B, int32_t H, int32_t Sq);

/* ------------------------------------------------------------------ the float32 flavour of the training op
 * The reference computes in `--dtype`, and its own default is fp32 (lwm/train.py:36; the launch scripts of
 * scripts/run_train_*.sh pass --dtype='fp32'; BASELINE configs[0] is the fp32 model).  These four entry points are
 * lwm_attn_fwd / lwm_attn_bwd_delta / _dq / _dkdv with FLOAT32 tensors behind every LwmTensor4 of LwmAttnArgs
 * (q, k, v, out, dout, dq, dk, dv: f32, D contiguous, strides in elements and multiples of 4; lse, the carries and the
 * row statistics are what they are in the bf16 flavour, and the statistics buffer has the same size).  Same mask
 * semantics, same carries (carry_in / final_out), same "rows with no visible key give 0 / -inf"; every contraction runs
 * on the exact-f32 matrix instruction (v_mfma_f32_32x32x2_f32), no operand is rounded to bf16.  Not taken:
 * piecewise position maps (q_pieces / k_pieces > 1), dense_mask, k_splits -> LWM_EUNSUPPORTED; the block-sparsity
 * hints are accepted and not read.  Kernels: lwm_amd/csrc/attn_f32.h. */
int lwm_attn_fwd_f32(const LwmAttnArgs* args, void* stream);
int lwm_attn_bwd_delta_f32(const LwmAttnArgs* args, void* stream);
int lwm_attn_bwd_dq_f32(const LwmAttnArgs* args, void* stream);
int lwm_attn_bwd_dkdv_f32(const LwmAttnArgs* args, void* stream);
/* The steps either side of the op at dtype = fp32 (lwm_amd/csrc/elem_f32.h): lwm_rope_bf16 / lwm_rmsnorm_{fwd,bwd}_bf16 /
 * lwm_swiglu_{fwd,bwd}_bf16 / lwm_softmax_ce_bf16 with float tensors (rows of C % 4 == 0, V % 4 == 0, n % 4 == 0, 16-byte
 * aligned), same arithmetic minus the roundings to bf16 (at dtype = f32 the reference's casts are identities,
 * lwm/llama.py:339-341); lwm_rmsnorm_bwd_f32 takes lwm_rmsnorm_bwd_workspace_bytes(rows, C) bytes of workspace.
 * lwm_sum_f32: dst = ((srcs[0] + srcs[1]) + ...) in argument order -- lwm_sum_f32_to_bf16 with a float result (the
 * owner-side reduction of returned dK / dV partials when the operands are f32). */
int lwm_rope_f32(LwmTensor4 x, LwmTensor4 y, const float* table, const int32_t* pos, int32_t B, int32_t S, int32_t H,
                 int32_t D, int32_t max_pos, int32_t conj, void* stream);
int lwm_rmsnorm_fwd_f32(const float* x, const float* w, float* y, float* rstd, int64_t rows, int32_t C, float eps,
                        void* stream);
int lwm_rmsnorm_bwd_f32(const float* x, const float* w, const float* g, const float* rstd, float* dx, float* dw,
                        void* workspace, int64_t rows, int32_t C, void* stream);
int lwm_swiglu_fwd_f32(const float* a, const float* b, float* y, int64_t n, void* stream);
int lwm_swiglu_bwd_f32(const float* a, const float* b, const float* g, float* da, float* db, int64_t n, void* stream);
int lwm_softmax_ce_f32(const float* logits, const int32_t* target, const float* weight, float* nll, int32_t* correct,
                       float* dlogits, int64_t rows, int32_t V, void* stream);
int lwm_sum_f32(const float* const* srcs, int32_t n_src, float* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------ the sequence ring
 * The exchange that lax.ppermute performs under ringattention (lwm/llama.py:539-569, SURVEY.md Appendix
 * A.1), driven from C: for ring step t rank r holds the K/V block of rank (r - t) mod n, runs the local
 * queries against it on the COMPUTE stream while the block travels on to rank r+1 and the next one arrives
 * from rank r-1 on the SIDE stream (grouped ncclSend / ncclRecv of RCCL over xGMI), hipEvents handing the
 * double buffer back and forth.  The backward rotates K, V and the f32 dK/dV carries of the block; after n
 * steps they are home.  Ownership is the reference's: rank r holds positions [r*c, (r+1)*c) (contiguous,
 * lwm/llama.py:560-562); attn_bias / segment_ids are replicated, full length (lwm/llama.py:563-564).
 *
 * A ring object holds no device memory: the caller passes a workspace of lwm_ring_workspace_bytes()
 * bytes (K/V double buffer, f32 carries, mask slices), which must stay untouched until the compute stream
 * has passed the call.  Transports:
 *   lwm_ring_create            an existing ncclComm_t of the "sp" group (RCCL is resolved at run time:
 *                              dlsym in the process, else dlopen librccl.so -- the library has no link-time
 *                              dependency on it);
 *   lwm_ring_create_from_id    the library creates (and owns) the communicator from an ncclUniqueId
 *                              (lwm_ring_unique_id on rank 0, broadcast by the host's own means);
 *   lwm_ring_create_transport  any send/recv pair given as function pointers (tests drive the schedule
 *                              through it with in-process mailboxes; n = 1 needs no transport at all).
 */
typedef struct LwmRing LwmRing;

typedef struct LwmRingTransport {
    void* ctx;
    int (*group_start)(void* ctx);
    /* enqueue on `stream` (a hipStream_t): send `bytes` bytes at `buf` to ring rank `peer` / receive from it */
    int (*send)(void* ctx, const void* buf, int64_t bytes, int32_t peer, void* stream);
    int (*recv)(void* ctx, void* buf, int64_t bytes, int32_t peer, void* stream);
    int (*group_end)(void* ctx);
} LwmRingTransport;

typedef struct LwmRingArgs {
    LwmTensor4 q, k, v;      /* local shards [B,c,H,D] bf16; k and v dense (they are sent as they are) */
    LwmTensor4 out;          /* fwd: written; bwd: read */
    float* lse;              /* [B,H,c] f32: fwd writes, bwd reads */
    LwmTensor4 dout;         /* bwd in */
    LwmTensor4 dq, dk, dv;   /* bwd out, [B,c,H,D] bf16 */
    const int32_t* segment_ids; /* [B,S_global] or NULL -- replicated, full length */
    const uint8_t* key_valid;   /* [B,S_global] or NULL */
    int32_t B, c, H, D;      /* c = local sequence length = S_global / n */
    float scale;
    int32_t causal;
    void* workspace;         /* lwm_ring_workspace_bytes(B, c, H, D, backward, n, schedule) bytes, 256-byte aligned */
    int32_t layout;          /* LWM_RING_LAYOUT_*: which positions a rank owns */
    int32_t schedule;        /* LWM_RING_SCHEDULE_*: how K/V and the dK/dV contributions move */
    /* LWM_RING_LAYOUT_TABLE: HOST array of n_chunks = n * P ints (1 <= P <= LWM_MAX_PIECES): the sequence is cut into
     * n_chunks equal chunks, chunk j (positions [j * S/n_chunks, (j+1) * S/n_chunks)) belongs to rank chunk_owner[j]; every
     * rank owns exactly P chunks and holds them in ascending position order in its local rows.  Ignored otherwise. */
    const int32_t* chunk_owner;
    int32_t n_chunks;
    /* Gathered form only (lwm_ring_last_form): a caller-owned device buffer of lwm_ring_kv_keep_bytes() bytes that the
     * FORWARD gathers the fetched K/V into (instead of its workspace) and that the BACKWARD of the same layer, handed the
     * same buffer with kv_kept = 1, reads instead of fetching K/V again: a quarter of the bytes a layer moves over xGMI
     * (K/V twice + f32 partials once -> K/V once + partials), for (n - 1) * c * H * D * 4 bytes per layer kept between the
     * two calls -- 0.5 GB at S = 32768 over 8 ranks, 1.9 GB at S = 131072; 288 GB per GPU is what makes that a choice.
     * EVERY rank of the ring must make the same choice for a call (a rank that does not fetch does not send either).
     * NULL / 0 = fetch in both calls.  Ignored by the per-pair form. */
    void* kv_keep;
    int32_t kv_kept;
} LwmRingArgs;

/* Ownership.  CONTIGUOUS = the reference's: rank r holds positions [r*c, (r+1)*c) (lwm/llama.py:560-562).
 * ZIGZAG: rank r holds the half-chunks r and 2n-1-r (c/2 positions each, local rows [0,c/2) and [c/2,c)) -- the
 * permutation applied at the boundary that balances causal work: under contiguous ownership rank n-1 computes n
 * blocks and rank 0 one.  lse is an OPAQUE residual between lwm_ring_attn_fwd and _bwd of the same ring and geometry
 * (two dense [B,H,c/2] pieces, one per segment, or one [B,H,c] piece in local row order, as the form of the call has it);
 * everything else keeps its [B,c,H,D] shape in local row order. */
/* TABLE: any ownership of equal chunks, P per rank, handed over as LwmRingArgs::chunk_owner -- for packed batches, where
 * zigzag balances a full causal triangle but not documents: at BASELINE configs[4] (1,048,576 tokens in 15 documents, 8
 * ranks) the slowest zigzag rank computes 1.9x the mean; four chunks per rank assigned by visible-pair count (a loader
 * knows the document lengths: lwm_amd/ring.py balanced_layout) bring that to ~1.03.  Runs in the gathered form only
 * (lwm_ring_last_form): direct schedule, B = 1, causal, chunks of a multiple of 256 rows. */
enum { LWM_RING_LAYOUT_CONTIGUOUS = 0, LWM_RING_LAYOUT_ZIGZAG = 1, LWM_RING_LAYOUT_TABLE = 2 };
/* Exchange.  RING = the reference's (lax.ppermute i -> i+1): the K/V block, and in the backward its f32 dK/dV
 * carry, hop to the next rank once per step -- every byte crosses ONE link per step, n-1 (n) times.
 * DIRECT: MI355X's xGMI is a full mesh, so nothing is forwarded: one grouped exchange brings every rank exactly
 * the K/V segments its queries can see from their owners (all 7 links busy at once; zigzag + causal: 1/4 fewer
 * bytes than rotating whole blocks), and in the backward the f32 dK/dV partial of each remote block goes straight
 * back to its owner, which sums the <= n partials in a fixed order (lwm_sum_f32_to_bf16).  Results agree with RING
 * up to the f32 association of that sum. */
enum { LWM_RING_SCHEDULE_RING = 0, LWM_RING_SCHEDULE_DIRECT = 1 };

int lwm_ring_create(void* nccl_comm, int32_t rank, int32_t n, void* side_stream, LwmRing** out);
int lwm_ring_unique_id(void* id128);   /* 128 bytes = ncclUniqueId */
int lwm_ring_create_from_id(const void* id128, int32_t rank, int32_t n, void* side_stream, LwmRing** out);
int lwm_ring_create_transport(const LwmRingTransport* transport, int32_t rank, int32_t n, void* side_stream,
                              LwmRing** out);
int lwm_ring_destroy(LwmRing* ring);
int64_t lwm_ring_workspace_bytes(int32_t B, int32_t c, int32_t H, int32_t D, int32_t backward, int32_t n,
                                 int32_t schedule);
int lwm_ring_attn_fwd(LwmRing* ring, const LwmRingArgs* args, void* compute_stream);
int lwm_ring_attn_bwd(LwmRing* ring, const LwmRingArgs* args, void* compute_stream);
/* What ONE call of lwm_ring_attn_fwd (backward = 0) or lwm_ring_attn_bwd (backward = 1) makes rank `rank` send, in
 * bytes, under the given ownership and schedule -- a pure function of the geometry (no device is touched): the
 * forward's K/V, and in the backward the K/V again plus the f32 dK/dV carries (ring) or partials (direct). */
int64_t lwm_ring_planned_bytes(int32_t layout, int32_t schedule, int32_t n, int32_t rank, int32_t B, int32_t c, int32_t H,
                               int32_t D, int32_t causal, int32_t backward);
/* The same for LWM_RING_LAYOUT_TABLE (direct schedule, causal, B = 1 -- the form a table runs in): a rank sends each of its
 * chunks to every peer whose last chunk lies above it and, in the backward, returns an f32 dK and dV partial for every
 * chunk it fetched.  -1 for an unusable table. */
int64_t lwm_ring_planned_bytes_table(const int32_t* chunk_owner, int32_t n_chunks, int32_t n, int32_t rank, int32_t c, int32_t H,
                                     int32_t D, int32_t backward);
/* Size of LwmRingArgs::kv_keep for a shard shape: room for the K and the V rows of every other rank, (n - 1) * c rows each. */
int64_t lwm_ring_kv_keep_bytes(int32_t B, int32_t c, int32_t H, int32_t D, int32_t n);
/* bytes this ring object has sent since creation (diagnostic) */
int64_t lwm_ring_bytes_sent(const LwmRing* ring);
/* Which form the last lwm_ring_attn_fwd / _bwd call took (diagnostic): 0 = one launch per (q segment, k segment) pair,
 * chained through f32 carries; 1 = the GATHERED form of the direct schedule -- the fetched K/V segments laid down in
 * position order in one buffer and read through two-piece position maps (LwmAttnArgs::q_split / k_split): per kernel
 * two launches per call, the local block under the fetch and everything that arrived after it.  Taken when the call is
 * causal, B = 1, the K/V fetch is one group (lwm_ring_set_fetch_groups: the default) and every segment of a shard is a
 * multiple of 256 rows; same results up to f32 association (fewer carries: closer to the exact sum). */
int lwm_ring_last_form(const LwmRing* ring);
/* Direct schedule: the K/V fetch of a call goes out as `groups` grouped exchanges over contiguous ranges of rank distance,
 * nearest first, each with its own arrival event; step t of the schedule waits only for the group that holds distance t.
 * 1 (the default: one bulk-synchronous group keeps all xGMI links busy, and the kernels of the remote blocks run as ONE
 * launch per kernel over everything that arrived -- lwm_ring_last_form) ... n - 1 (every block is handed over as it lands, a
 * launch per block: for transports whose transfers are serial anyway, such as the IPC one).  Values above n - 1 mean n - 1. */
int lwm_ring_set_fetch_groups(LwmRing* ring, int32_t groups);
/* Diagnostic (rings created with LWM_RING_TIMING=1 in the environment: their events then carry timestamps): after the
 * last lwm_ring_attn_fwd / _bwd call has completed, the milliseconds from the call's entry to (kv_ms[t]) the arrival of
 * the K/V block of rank distance t (direct schedule, t >= 1) and to (kernels_ms[t]) the end of the kernels of step t;
 * count >= n entries each, -1 where undefined.  Blocks on the call's completion. */
int lwm_ring_fetch_timeline(LwmRing* ring, float* kv_ms, float* kernels_ms, int32_t count);
/* Diagnostic: ONE grouped exchange with ourselves through the ring's transport -- send `bytes` bytes at src to
 * our own ring rank and receive them into dst -- enqueued on the ring's side stream and handed over by the same
 * events an attention step uses (compute stream -> side stream -> compute stream).  With a ring made by
 * lwm_ring_create_from_id(n = 1) this is a complete first contact with RCCL on a single GPU: the run-time symbol
 * table, the by-value ncclUniqueId, ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a non-default stream. */
int lwm_ring_selftest(LwmRing* ring, const void* src, void* dst, int64_t bytes, void* compute_stream);

/* A transport that needs neither RCCL nor compute units: every rank owns a mailbox (device memory) that its peers map
 * through hipIpcMemHandles; a message is one hipMemcpyAsync into the receiver's mailbox (SDMA over xGMI between GPUs)
 * followed by a stream memory operation on an uncached flag word (hipStreamWriteValue32 / hipStreamWaitValue32), and
 * one local copy out of the mailbox on the receiving side.  All 256 CUs stay with the attention kernels, and because
 * IPC also works between processes sharing ONE GPU the complete multi-process driver can run on a single-GPU box.
 *   1. every rank: lwm_ring_ipc_export(rank, n, slot_bytes, slots, info, &ipc) -- allocates mailbox + flags on the
 *      current device; info receives lwm_ring_ipc_info_bytes() bytes to publish.  slot_bytes >= the largest message
 *      (a K/V or f32 dK/dV piece: B * c * H * D * 4 bytes covers everything the driver sends), slots >= 4 * B
 *      messages per ordered pair and group (8 is a good default); memory = n * slots * slot_bytes per rank.
 *   2. the host all-gathers the n info blobs by its own means (rank-major) -> lwm_ring_ipc_connect(ipc, all_infos).
 *   3. lwm_ring_create_ipc(ipc, side_stream, &ring) -- a ring object like any other; destroy the ring, then the ipc. */
typedef struct LwmRingIpc LwmRingIpc;
int64_t lwm_ring_ipc_info_bytes(void);
int lwm_ring_ipc_export(int32_t rank, int32_t n, int64_t slot_bytes, int32_t slots, void* info_out, LwmRingIpc** out);
int lwm_ring_ipc_connect(LwmRingIpc* ipc, const void* all_infos);
int lwm_ring_create_ipc(LwmRingIpc* ipc, void* side_stream, LwmRing** out);
int lwm_ring_ipc_destroy(LwmRingIpc* ipc);

/* (min, max) of segment_ids over each block of 32 rows, excluding rows whose valid[] is 0
 * (valid may be NULL); an all-invalid block gets (INT32_MAX, INT32_MIN).
 * blocks: [B][ceil(S/32)][2] int32. */
int lwm_attn_segment_blocks(const int32_t* segment_ids, const uint8_t* valid, int32_t* blocks,
                            int32_t B, int32_t S, void* stream);

/* Merge P normalised partial attention results (split-K pieces of one launch, or
 * the per-rank partials of a sequence-sharded K/V cache, lwm/llama.py:599-614):
 *   o_parts [P][B,Sq,H,D] f32, lse_parts [P][B,H,Sq] f32  ->
 *   out bf16 [B,Sq,H,D] (strided) if out.ptr != NULL, else out_f32 [B,Sq,H,D];
 *   lse [B,H,Sq] f32 (may be NULL).  Rows whose every partial is empty
 *   (lse = -inf) give out = 0, lse = -inf. */
int lwm_attn_combine(const float* o_parts, const float* lse_parts, int32_t P, LwmTensor4 out,
                     float* out_f32, float* lse, int32_t B, int32_t Sq, int32_t H, int32_t D,
                     void* stream);

/* KV-cache write (lwm/llama.py:440-492): cache[b, dst_row0 + i, :] = src[b, src_row0 + i, :]
 * for i < nrows; rows are row_elems bf16 (= H*D) contiguous; batch strides in
 * elements.  The caller decides which rows land in its shard (decode: only the
 * owning sp shard writes, :454-467). */
int lwm_kv_cache_write(void* cache, const void* src, int32_t B, int64_t cache_stride_b,
                       int64_t src_stride_b, int64_t dst_row0, int64_t src_row0, int64_t nrows,
                       int32_t row_elems, void* stream);

/* The same with the destination row read from DEVICE memory: row = *dst_row0_dev + row_offset + i;
 * rows outside [0, cache_rows) are skipped, which is the decode rule "only the owning sp shard
 * writes" (lwm/llama.py:454-467) with row_offset = -rank * cache_rows.  No host value changes from
 * one decode step to the next, so a step can be captured in a hipGraph and replayed. */
int lwm_kv_cache_write_at(void* cache, const void* src, int32_t B, int64_t cache_stride_b,
                          int64_t src_stride_b, const int32_t* dst_row0_dev, int64_t row_offset,
                          int64_t cache_rows, int64_t src_row0, int64_t nrows, int32_t row_elems,
                          void* stream);

/* Elementwise helpers of the ring driver (HBM-bound). */
/* dst_bf16[n] = (bf16) src_f32[n] */
int lwm_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst_bf16[n] = (bf16) (((srcs[0] + srcs[1]) + srcs[2]) + ...), f32 adds in argument order.
 * `srcs` is a HOST array of n_src (1..16) device pointers.  Replaces the f32 dk/dv carry
 * that the reference's backward scan ppermutes around the ring:
 * under the mesh schedule every rank returns its partial straight to the block's owner,
 * which reduces them here. */
int lwm_sum_f32_to_bf16(const float* const* srcs, int32_t n_src, void* dst, int64_t n, void* stream);

/* ------------------------------------------------------------------ RoPE, RMSNorm
 * The HBM-bound steps either side of the attention op. */

/* apply_rotary_emb (lwm/llama.py:353-375): y = x rotated, interleaved (even, odd) pairs as
 * complex numbers times (cos, sin)[pos[b,s]][i]; f32 math, bf16 in/out; y may alias x.
 * table: [max_pos][D/2][2] f32 built on the host as precompute_freqs_cis does
 * (lwm/llama.py:344-350).  conj != 0 rotates by the negative angle (= backward). */
int lwm_rope_bf16(LwmTensor4 x, LwmTensor4 y, const float* table, const int32_t* pos, int32_t B,
                  int32_t S, int32_t H, int32_t D, int32_t max_pos, int32_t conj, void* stream);

/* RMSNorm (lwm/llama.py:320-341) over rows of C bf16: y = bf16(bf16(x * rsqrt(mean(x^2)+eps)) * w).
 * rstd [rows] f32 is written if non-NULL (needed by the backward). */
int lwm_rmsnorm_fwd_bf16(const void* x, const void* w, void* y, float* rstd, int64_t rows, int32_t C,
                         float eps, void* stream);
/* dx [rows,C] bf16 and dw [C] bf16 from g = dL/dy; workspace of
 * lwm_rmsnorm_bwd_workspace_bytes() bytes (f32 partial dW per workgroup). */
int64_t lwm_rmsnorm_bwd_workspace_bytes(int64_t rows, int32_t C);
int lwm_rmsnorm_bwd_bf16(const void* x, const void* w, const void* g, const float* rstd, void* dx,
                         void* dw, void* workspace, int64_t rows, int32_t C, void* stream);
/* The same with the gradient of the residual branch that by-passes the norm folded in (FlaxLLaMABlock,
 * lwm/llama.py:704-744: `x` feeds the norm AND the residual add behind it): dx = bf16(bf16(dx_norm) + res) -- the
 * roundings of a separate bf16 add, one pass less over (rows, C).  res == NULL: lwm_rmsnorm_bwd_bf16. */
int lwm_rmsnorm_bwd_res_bf16(const void* x, const void* w, const void* g, const float* rstd, const void* res,
                             void* dx, void* dw, void* workspace, int64_t rows, int32_t C, void* stream);

/* The SwiGLU gate of FlaxLLaMAMLP (lwm/llama.py:659): y = silu(a) * b and its backward
 * (da, db from g = dL/dy); bf16, n % 8 == 0, all pointers 16-byte aligned. */
int lwm_swiglu_fwd_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);
int lwm_swiglu_bwd_bf16(const void* a, const void* b, const void* g, void* da, void* db, int64_t n,
                        void* stream);
/* The same on (rows, cols) windows of wider buffers (leading dimensions in elements, multiples of 8): gate and up as the
 * two halves of ONE (rows, 2F) GEMM output -- w1 | w3 run as one library GEMM (lwm/llama.py:631-655 share their input) --
 * and d gate | d up written into the halves of the one buffer the fused dgrad / wgrad GEMMs read. */
int lwm_swiglu_fwd_ld_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, void* y, int64_t ldy, int64_t rows,
                           int64_t cols, void* stream);
int lwm_swiglu_bwd_ld_bf16(const void* a, int64_t lda, const void* b, int64_t ldb, const void* g, int64_t ldg, void* da,
                           int64_t ldda, void* db, int64_t lddb, int64_t rows, int64_t cols, void* stream);

/* dst[c][r] = src[r][c], bf16, rows and cols multiples of 64 (leading dimensions in elements).  Serves the library GEMMs
 * around the hot path: flax Dense kernels are (in, out) (lwm/llama.py:390-421, :631-655); hipBLASLt runs fastest with the
 * reduction dimension contiguous in both operands, so the harness re-lays each kernel as (out, in) once per step and
 * transposes the narrow operand of each weight gradient.  HBM-bound (2 x rows x cols x 2 bytes). */
int lwm_transpose_bf16(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                       void* stream);

/* The weight gradient of a flax Dense kernel (lwm/llama.py:390-421, :631-655: y = x @ W with W (in, out)):
 * dw[K][N] = sum over s of x[s][K] * g[s][N], bf16 operands with the feature dimension contiguous (leading dimensions in
 * elements), f32 accumulation in a fixed order, bf16 result.  S % 32 == 0, K % 256 == 0, N % 256 == 0.  Both operands are
 * read where they lie (no transposed copies): LDS-DMA tiles, transposed MFMA fragments.  Tiles beyond the last whole round
 * of one tile per CU are cut along S (stream-K) and summed from f32 partials in `workspace`
 * (lwm_wgrad_workspace_bytes(S, K, N), 16-byte aligned; may be null when that is 0).
 * FLOPs 2 S K N; HBM bytes 2 (S K + S N + K N) algorithmic. */
int64_t lwm_wgrad_workspace_bytes(int64_t S, int64_t K, int64_t N);
int lwm_wgrad_bf16(const void* x, int64_t ldx, const void* g, int64_t ldg, void* dw, int64_t lddw, int64_t S, int64_t K,
                   int64_t N, void* workspace, int64_t workspace_bytes, void* stream);

/* y[r, :] = x[r, :] . W for rows <= 4 -- the projections of a cached-decode step (one token per batch row
 * against the [K, N] bf16 kernels wq/wk/wv/wo, w1/w2/w3, lm_head; x @ kernel as flax nn.Dense computes it,
 * lwm/llama.py:427-432, :659, :1075-1106).  HBM-bound: W is read once (2*K*N bytes), f32 accumulation in a fixed
 * order (deterministic).  x: [rows, K] bf16, row stride ldx; w: [K, N] bf16 dense; y: [rows, N] bf16 (row
 * stride ldy) and / or y_f32: [rows, N] f32 dense; workspace: lwm_gemv_workspace_bytes() bytes, 16-byte aligned.
 * N % 8 == 0, K % 32 == 0, K <= 12288.
 * lwm_gemv_multi_bf16: 1..3 kernels that share x (wq | wk | wv; w1 | w3) in ONE pair of launches; w, y, ldy,
 * y_f32, N are arrays of nmat entries (y or y_f32 may be NULL as a whole or per entry); workspace: the sum of
 * lwm_gemv_workspace_bytes(rows, K, N[i]). */
int64_t lwm_gemv_workspace_bytes(int32_t rows, int32_t K, int32_t N);
int lwm_gemv_bf16(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, float* y_f32, void* workspace,
                  int32_t rows, int32_t K, int32_t N, void* stream);
int lwm_gemv_multi_bf16(const void* x, int64_t ldx, int32_t nmat, const void* const* w, void* const* y, const int64_t* ldy,
                        float* const* y_f32, const int32_t* N, void* workspace, int32_t rows, int32_t K, void* stream);

/* The same launch pair with the small launches either side of a decode-step projection riding along (every field but
 * the first block optional, zero = off):
 *   RMSNorm on load  (lwm/llama.py:320-341 in front of wq|wk|wv, w1|w3, lm_head): x is normalised as it is read,
 *     bf16(bf16(x * rstd) * norm_weight[k]) with rstd = 1/sqrt(sum(ss_in[r][0..ss_n)) / K + eps) -- ss_in are partial
 *     sums of squares of x's rows, e.g. the ss_out of the launch that produced x (or one total and zeros);
 *   residual add     (lwm/llama.py:719, :737 behind wo, w2): y = bf16(bf16(x . W) + residual), and
 *   ss_out           [rows][N/128] partial sums of squares of y's rows (one matrix, N % 128 == 0) for the next norm.
 * Results equal the separate launches' (same roundings); rstd may differ in its last bit (other summation order). */
typedef struct LwmGemvArgs {
    const void* x; int64_t ldx; int32_t nmat; int32_t rows, K;
    const void* w[3]; void* y[3]; int64_t ldy[3]; float* y_f32[3]; int32_t N[3];
    void* workspace;
    const void* norm_weight; const float* ss_in; int32_t ss_n; float eps;
    const void* residual[3]; int64_t ldres[3];
    float* ss_out;
} LwmGemvArgs;
int lwm_gemv_fused_bf16(const LwmGemvArgs* args, void* stream);

/* tux.cross_entropy_loss_and_accuracy as used at lwm/train.py:177-181, :192-201, per row of
 * bf16 logits [rows, V] (V % 8 == 0, V <= 32768): nll[r] = logsumexp(row) - row[target[r]] in
 * f32; correct[r] = (first argmax == target[r]) (may be NULL); and, if dlogits != NULL, the
 * fused gradient dlogits[r] = (softmax(row) - onehot(target[r])) * weight[r] (weight NULL = 1).
 * dlogits may alias logits. */
int lwm_softmax_ce_bf16(const void* logits, const int32_t* target, const float* weight, float* nll,
                        int32_t* correct, void* dlogits, int64_t rows, int32_t V, void* stream);

/* ------------------------------------------------------------------ VQGAN
 * Primitives of the video tokeniser, lwm/vqgan.py.  All tensors are f32, NHWC,
 * dense; results are bit-exact with oracle/vqgan_ref.c (exact-f32 MFMA, fixed
 * summation order -- see DESIGN.md "VQGAN arithmetic contract").
 */

/* flax nn.Conv (kernel HWIO [KH,KW,Cin,Cout], bias) on x [B,Hin,Win,Cin] ->
 * y [B,Ho,Wo,Cout].  Output (oy,ox), tap (kh,kw) reads the virtual input
 * (x upsampled nearest by 2^up_shift) at (oy*stride+kh-pad, ox*stride+kw-pad),
 * zero outside.  Replaces:
 *   3x3 SAME  (stride 1, pad 1)                     lwm/vqgan.py:155,163,172-175,183,253,257
 *   1x1       (stride 1, pad 0)                     lwm/vqgan.py:114-115,262
 *   Downsample (stride 2, pad 0, Ho=Hin/2: the jnp.pad of one zero row/column at
 *              bottom/right is the zero fill)       lwm/vqgan.py:286-303
 *   Upsample  (up_shift 1, stride 1, pad 1)         lwm/vqgan.py:306-319
 * residual (same shape as y, may be NULL) is added after the bias
 * (ResnetBlock, lwm/vqgan.py:263); clip != 0 clamps to [-1,1] (lwm/vqgan.py:141). */
typedef struct LwmConvArgs {
    const float* x;
    const float* w;
    const float* bias;     /* [Cout] or NULL */
    const float* residual; /* [B,Ho,Wo,Cout] or NULL */
    float* y;
    int32_t B, Hin, Win, Cin, Cout, KH, KW, stride, pad, up_shift, Ho, Wo, clip;
} LwmConvArgs;
int lwm_conv2d_nhwc_f32(const LwmConvArgs* args, void* stream);

/* flax nn.GroupNorm (num_groups G, eps, affine) over x [B,HW,C], optionally
 * followed by nn.silu (lwm/vqgan.py:161-162,181-182,251-255).  `workspace` is
 * caller-owned device memory of lwm_groupnorm_workspace_bytes() bytes (f64
 * slice partials, then the f32 mean / rstd per image and group); y may alias x. */
int64_t lwm_groupnorm_workspace_bytes(int32_t B, int64_t HW, int32_t C, int32_t G);
int lwm_groupnorm_silu_f32(const float* x, const float* gamma, const float* beta, float* y,
                           void* workspace, int32_t B, int64_t HW, int32_t C, int32_t G, float eps,
                           int32_t silu, void* stream);

/* VectorQuantizer (lwm/vqgan.py:187-221), D = 64.
 * lwm_vq_sqnorm_f32 : se[e] = |codebook[e]|^2 (once per codebook)
 * lwm_vq_argmin_f32 : idx[n] = first argmin_e (|z_n|^2 + se[e]) - 2 z_n.e   (:207-212)
 * lwm_vq_gather_f32 : out[n] = codebook[idx[n]] (z NULL; decode path :204-205) or the
 *                     forward value of z + stop_gradient(z_q - z) (:214) */
int lwm_vq_sqnorm_f32(const float* codebook, float* se, int32_t E, int32_t D, void* stream);
int lwm_vq_argmin_f32(const float* z, const float* codebook, const float* se, int32_t* idx,
                      int64_t N, int32_t E, int32_t D, void* stream);
int lwm_vq_gather_f32(const float* codebook, const int32_t* idx, const float* z, float* out,
                      int64_t N, int32_t E, int32_t D, void* stream);

const char* lwm_last_error(void);
int lwm_version(void);
/* sizeof(LwmAttnArgs) (which = 0) / sizeof(LwmConvArgs) (1) / sizeof(LwmRingArgs) (2) as compiled into the library:
 * lets a foreign-language binding verify its struct mirror at load time. */
int lwm_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif /* LWM_HIP_H */
