"""`--mesh_dim` of the reference's entry points (lwm/train.py:35, lwm/vision_chat.py:26: e.g. '1,-1,1,1',
'!1,1,4,8', 'dp:1,fsdp:1,tp:1,sp:8') and the process groups it implies.

The reference builds a jax Mesh with axes ('dp', 'fsdp', 'tp', 'sp') (lwm/llama.py:201-203, tux.get_jax_mesh)
and shards the sequence over 'sp'.  Here ranks take the place of devices: rank = ((dp*F + fsdp)*T + tp)*S + sp,
'sp' fastest -- consecutive ranks of one node form an sp ring, so the K/V exchange stays on xGMI -- and
`sp_group()` returns the torch.distributed group this rank's RingAttention runs over
(`lwm_amd.ringattention.set_sp_group`)."""
from __future__ import annotations

AXES = ("dp", "fsdp", "tp", "sp")


def parse_mesh_dim(text: str, world_size: int) -> dict:
    """-> {'dp': d, 'fsdp': f, 'tp': t, 'sp': s} with d*f*t*s == world_size; one entry may be -1."""
    text = text.strip()
    if text.startswith("!"):          # tux: mesh-axis splitting flag; no meaning for one process per GPU
        text = text[1:]
    parts = [p.strip() for p in text.split(",") if p.strip()]
    if any(":" in p for p in parts):
        named = dict(p.split(":", 1) for p in parts)
        if set(named) != set(AXES):
            raise ValueError(f"mesh_dim must name exactly {AXES}, got {sorted(named)}")
        dims = [int(named[a]) for a in AXES]
    else:
        if len(parts) != len(AXES):
            raise ValueError(f"mesh_dim needs {len(AXES)} comma-separated sizes (dp,fsdp,tp,sp), got {text!r}")
        dims = [int(p) for p in parts]
    if dims.count(-1) > 1 or any(d == 0 or d < -1 for d in dims):
        raise ValueError(f"bad mesh_dim {text!r}")
    known = 1
    for d in dims:
        if d != -1:
            known *= d
    if -1 in dims:
        if world_size % known:
            raise ValueError(f"mesh_dim {text!r} does not divide {world_size} ranks")
        dims[dims.index(-1)] = world_size // known
    elif known != world_size:
        raise ValueError(f"mesh_dim {text!r} describes {known} ranks, the job has {world_size}")
    return dict(zip(AXES, dims))


def coords(mesh: dict, rank: int) -> dict:
    out = {}
    for a in reversed(AXES):
        out[a] = rank % mesh[a]
        rank //= mesh[a]
    return out


def axis_ranks(mesh: dict, rank: int, axis: str) -> list:
    """The ranks that differ from `rank` only along `axis`, in axis order."""
    strides, s = {}, 1
    for a in reversed(AXES):
        strides[a] = s
        s *= mesh[a]
    base = rank - coords(mesh, rank)[axis] * strides[axis]
    return [base + i * strides[axis] for i in range(mesh[axis])]


def sp_group(mesh: dict, backend=None):
    """Creates every 'sp' group (new_group is collective: all ranks create all groups, in the same order)
    and returns this rank's.  With sp == world size this is the default group."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if mesh["sp"] == world:
        return dist.group.WORLD
    mine, seen = None, set()
    for r in range(world):
        ranks = tuple(axis_ranks(mesh, r, "sp"))
        if ranks in seen:
            continue
        seen.add(ranks)
        g = dist.new_group(list(ranks), backend=backend)
        if rank in ranks:
            mine = g
    return mine
