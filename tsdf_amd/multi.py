"""Z-slab sharding of one TSDF volume over the GPUs of a node (one process per GPU, torch.distributed).

Integrate shards with no communication: voxels are independent, so rank r integrates planes
[z_begin, z_end) plus one halo plane above (recomputed locally, deterministic).  Raycast has one exchange
step: every rank marches the GLOBAL sample lattice of every ray but evaluates only the samples whose lower
trilinear tap plane it owns, records the first owned sample with tsdf <= 0 as {k, x, y, z}, and a single
all-gather of those 16-byte records (RCCL over xGMI: W*H*16 B = 4.9 MB per rank) followed by a per-pixel
min-k select reproduces the single-GPU vertex map bit for bit.
"""
import torch
import torch.distributed as dist


def slab_range(size_z, world, rank):
    """Planes [z_begin, z_end) owned by `rank`: contiguous, sizes differ by at most one plane."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    if world > size_z:
        raise ValueError("cannot split %d planes over %d ranks" % (size_z, world))
    base, rem = divmod(size_z, world)
    z_begin = rank * base + min(rank, rem)
    return z_begin, z_begin + base + (1 if rank < rem else 0)


def resident_range(size_z, world, rank):
    """Planes a rank keeps in HBM: its slab plus one halo plane above (trilinear reads lower.z + 1)."""
    zb, ze = slab_range(size_z, world, rank)
    return zb, min(ze + 1, size_z)


def owner_of_plane(size_z, world, z):
    for r in range(world):
        zb, ze = slab_range(size_z, world, r)
        if zb <= z < ze:
            return r
    raise ValueError("plane %d outside 0..%d" % (z, size_z))


def gather_hits(hits_mine, hits_all=None, group=None):
    """All-gather of the per-pixel hit records: (W*H, 4) float32 per rank -> (world, W*H, 4)."""
    world = dist.get_world_size(group)
    if hits_all is None:
        hits_all = torch.empty((world,) + tuple(hits_mine.shape), dtype=hits_mine.dtype, device=hits_mine.device)
    dist.all_gather_into_tensor(hits_all.view(-1), hits_mine.contiguous().view(-1), group=group)
    return hits_all


def merge_hits(hits_all, width, height, vertices=None, stream=None):
    """Per-pixel min-k select on the GPU (tsdf_merge_hits_device).  CUDA tensors only: there is no CPU path."""
    from .api import merge_hits_device
    if not hits_all.is_cuda:
        raise TypeError("merge_hits runs the HIP merge kernel and needs CUDA tensors")
    if vertices is None:
        vertices = torch.empty((width * height, 3), dtype=torch.float32, device=hits_all.device)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    merge_hits_device(hits_all.data_ptr(), hits_all.shape[0], width, height, vertices.data_ptr(), s)
    return vertices


def gather_vertices(vertices_mine, group=None):
    """extract_surface over the slabs: every rank's (n_r, 3) float32 vertices (TSDFVolume.extract_surface on its slab)
    -> the whole volume's array on every rank, the ranks' parts in rank order (= the reference's cube order, z-major).
    The parts differ in length: lengths first, then one all-gather of the parts padded to the longest."""
    world = dist.get_world_size(group)
    mine = torch.as_tensor(vertices_mine, dtype=torch.float32).reshape(-1, 3)
    dev = mine.device
    n_mine = torch.tensor([mine.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.empty((world,), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n_mine, group=group)
    longest = int(counts.max().item())
    padded = torch.zeros((longest, 3), dtype=torch.float32, device=dev)
    padded[:mine.shape[0]] = mine
    parts = torch.empty((world, longest, 3), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(parts.view(-1), padded.view(-1), group=group)
    return torch.cat([parts[r, :int(counts[r].item())] for r in range(world)], dim=0)
