"""Z-slab sharding of one TSDF volume over the GPUs of a node (one process per GPU, torch.distributed).

Integrate shards with no communication: voxels are independent, so rank r integrates planes
[z_begin, z_end) plus one halo plane above (recomputed locally, deterministic).  Raycast has one exchange
step: every rank marches the GLOBAL sample lattice of every ray but evaluates only the samples whose lower
trilinear tap plane it owns, records the first owned sample with tsdf <= 0 as {k, t} (its index and its refined ray
parameter), and a single all-gather of those 8-byte records (RCCL over xGMI: W*H*8 B = 2.5 MB per rank) followed by a
per-pixel min-k select -- which forms the vertex from t and the pixel's own ray, identical on every rank -- reproduces the
single-GPU vertex map bit for bit.  (Up to round 2 the record carried the vertex: 16 bytes.)
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


def balanced_slab_ranges(cost_per_plane, world, min_planes=1):
    """Contiguous Z ranges [(z_begin, z_end)] * world that minimise the largest slab cost (still north_star's Z-slab split,
    only the boundaries move).  cost_per_plane: non-negative work estimate per plane (len = size_z).  Exact: the smallest
    feasible bound is found by bisection over the greedy 'fill until the bound' test, then the slabs are laid out left to
    right keeping at least `min_planes` planes for every slab still to come."""
    cost = [max(0.0, float(c)) for c in cost_per_plane]
    Z = len(cost)
    if min_planes < 1:
        raise ValueError("min_planes must be at least 1 (a slab volume holds at least one plane)")
    if world < 1 or world * min_planes > Z:
        raise ValueError("cannot split %d planes over %d ranks" % (Z, world))
    if world == 1:
        return [(0, Z)]
    total = sum(cost)
    if total <= 0.0:
        return [slab_range(Z, world, r) for r in range(world)]

    def layout(bound):
        """Greedy ranges under `bound` (None if more than `world` slabs are needed)."""
        ranges, z = [], 0
        for r in range(world):
            left = world - r - 1                      # slabs still to come
            end, acc = z, 0.0
            while end < Z - left * min_planes and (end - z < min_planes or acc + cost[end] <= bound):
                acc += cost[end]
                end += 1
            if r == world - 1 and end < Z:
                return None
            ranges.append((z, end))
            z = end
        return ranges if z == Z else None

    lo, hi = max(cost), total
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if layout(mid) is None:
            lo = mid
        else:
            hi = mid
    return layout(hi)


def refine_slab_ranges(ranges, seconds, size_z, min_planes=1, floor=None):
    """One round of measured rebalancing: `seconds[r]` is what rank r's slab `ranges[r]` cost for the same frames (integrate +
    slab ray cast).  A rank's time above the common floor (what even an empty slab costs: launches, ray set-up; default = 0.8 x
    the cheapest rank) is spread evenly over its planes, and balanced_slab_ranges re-cuts that density.  Every rank calls this
    with the same all-gathered numbers and gets the same plan."""
    world = len(ranges)
    t = [max(0.0, float(x)) for x in seconds]
    base = 0.8 * min(t) if floor is None else float(floor)
    density = [0.0] * size_z
    for (zb, ze), tr in zip(ranges, t):
        d = max(tr - base, 0.02 * max(t)) / max(1, ze - zb)
        for z in range(zb, ze):
            density[z] = d
    return balanced_slab_ranges(density, world, min_planes)


def plane_costs(volume_factory, frames, cameras, size, planner_z=128, integrate_weight=0.4, raycast_weight=0.6, constant=0.02,
                width=640, height=480):
    """Work estimate per Z plane of a `size` grid, for balanced_slab_ranges: a small planner volume of the same physical box
    (planner_z planes) integrates the given frames; per planner plane, the share of updated voxels (what integrate_kernel
    does per frame) and of occupied ray-caster bricks (where the march evaluates samples instead of jumping) are blended with
    the two kernels' share of a step, plus a constant per plane (cull, projection of voxels not updated).  Deterministic:
    every rank computes the same costs from the same frames, no communication.  volume_factory(grid_xyz) -> TSDFVolume."""
    import numpy as np
    X, Y, Z = (int(v) for v in size)
    pz = min(int(planner_z), Z)
    grid = (max(8, X * pz // Z), max(8, Y * pz // Z), pz)
    vol = volume_factory(grid)
    for depth, cam in zip(frames, cameras):
        vol.integrate(depth, int(width), int(height), cam)
    w = vol.get_weight_data().reshape(pz, -1)
    updated = (w > 0).sum(axis=1).astype(np.float64)
    fine = vol.occupancy_data(force_rebuild=True)[0]                     # (nbz, nby, nbx), 4-voxel bricks
    interior = fine[:, 1:-1, 1:-1] if fine.shape[1] > 2 and fine.shape[2] > 2 else fine
    occ = np.repeat(interior.reshape(interior.shape[0], -1).sum(axis=1).astype(np.float64), 4)[:pz]
    if hasattr(vol, "close"):
        vol.close()
    u = updated / updated.sum() if updated.sum() > 0 else np.full(pz, 1.0 / pz)
    o = occ / occ.sum() if occ.sum() > 0 else np.full(pz, 1.0 / pz)
    cost_p = integrate_weight * u + raycast_weight * o + constant / pz
    # planner planes -> grid planes
    idx = (np.arange(Z) * pz) // Z
    return (cost_p[idx] * pz / Z).tolist()


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


def new_hit_records(n_slabs, width, height, device="cuda"):
    """(n_slabs, W*H, 2) words: the records {uint32 k, float t} of n_slabs slabs (struct tsdf_hit_record), as float32 storage."""
    return torch.empty((n_slabs, width * height, 2), dtype=torch.float32, device=device)


def gather_hits(hits_mine, hits_all=None, group=None):
    """All-gather of the per-pixel hit records: (W*H, 2) words per rank -> (world, W*H, 2)."""
    world = dist.get_world_size(group)
    if hits_all is None:
        hits_all = torch.empty((world,) + tuple(hits_mine.shape), dtype=hits_mine.dtype, device=hits_mine.device)
    dist.all_gather_into_tensor(hits_all.view(-1), hits_mine.contiguous().view(-1), group=group)
    return hits_all


def merge_hits(volume, hits_all, width, height, camera, vertices=None, stream=None):
    """Per-pixel min-k select on the GPU (tsdf_merge_hits_device).  CUDA tensors only: there is no CPU path."""
    from .api import merge_hits_device
    if not hits_all.is_cuda:
        raise TypeError("merge_hits runs the HIP merge kernel and needs CUDA tensors")
    if vertices is None:
        vertices = torch.empty((width * height, 3), dtype=torch.float32, device=hits_all.device)
    s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
    merge_hits_device(volume, hits_all.data_ptr(), hits_all.shape[0], width, height, camera, vertices.data_ptr(), s)
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


class _DeviceWords:
    """A device buffer of 32-bit words as an object torch.as_tensor understands (__cuda_array_interface__)."""

    def __init__(self, ptr, words):
        self.__cuda_array_interface__ = {"shape": (int(words),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def device_words(ptr, words):
    """float32 tensor view of `words` 32-bit words of device memory at `ptr` (no copy; the caller keeps the memory alive)."""
    return torch.as_tensor(_DeviceWords(ptr, words), device="cuda")


class SlabExchange:
    """The frame's one collective (tsdf_slab_exchange, tsdf_amd/csrc/pipeline.hip), enqueued by RCCL **on the caller's HIP
    stream**: ncclAllGather of librccl.so called directly by the C++ library, with a communicator of its own whose unique id rank 0
    hands to the others through torch.distributed.  torch's own all_gather_into_tensor runs on the process group's internal
    stream and hands over to the caller's stream with events on either side; between kernels of that stream one call cost
    0.2-1.2 ms on a MI355X box (12.8 us back to back, tools/dbg_allgather.py) -- as much as the whole ray cast.  On the stream
    it is one more launch between the slab ray cast and the merge kernel and needs no synchronisation at all.

    SlabExchange(group)                -- RCCL (the librccl torch's nccl backend loaded), id broadcast through `group`
    SlabExchange(group, callback=fn)   -- fn(mine, all, stream_ptr): the caller's own collective on float32 views of the
                                          record buffers ((n_pixels * 2,), (world * n_pixels * 2,)); used for gloo / torch
                                          collectives in tests and on a box without direct RCCL
    all_gather(send, recv, stream): send = this rank's contiguous CUDA tensor of records, recv = (world, ...) of the same dtype."""

    def __init__(self, group=None, callback=None):
        import ctypes as C
        import os
        from ._capi import EXCHANGE_FN, check, lib
        self._C, self._lib, self._check = C, lib, check
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self._h = C.c_void_p()
        self._cb = None
        if callback is not None:
            def thunk(user, mine, allr, n_pixels, stream):
                try:
                    callback(device_words(mine, 2 * n_pixels), device_words(allr, 2 * n_pixels * self.world), stream or 0)
                    return 0
                except Exception:      # (an exception cannot cross the C frames)
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = EXCHANGE_FN(thunk)
            check(lib.tsdf_slab_exchange_create_callback(self.rank, self.world, self._cb, None, C.byref(self._h)))
            return
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so").encode()
        uid = (C.c_uint8 * 128)()
        # Every rank must leave this constructor the same way -- with a communicator or with an exception -- or the caller's fallback
        # (bench.py: torch's collective) would be taken by some ranks only and the next collective would hang.  So rank 0's failure to
        # make an id travels with the id (a status byte in front), and the ranks agree on the outcome of the communicator's creation.
        id_error = None
        if self.rank == 0:
            try:
                check(lib.tsdf_slab_exchange_unique_id(uid, path))
            except Exception as e_:
                id_error = e_
        # the id travels as 1 + 128 bytes through the process group that exists already (nccl: on the device; gloo: on the host)
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.frombuffer(bytearray(bytes([0 if id_error else 1]) + bytes(uid)), dtype=torch.uint8).clone().to(dev)
        dist.broadcast(t, src=0, group=group)
        got = bytes(t.cpu().numpy().tobytes())
        if got[0] == 0:
            raise RuntimeError("rank 0 could not make an RCCL id%s" % (": %s" % id_error if id_error else ""))
        C.memmove(uid, got[1:], 128)
        created = None
        try:
            check(lib.tsdf_slab_exchange_create(self.rank, self.world, uid, path, C.byref(self._h)))
        except Exception as e_:
            created = e_
        agree = torch.tensor([0 if created else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN, group=group)
        if int(agree.item()) == 0:
            self.close()
            raise RuntimeError("the RCCL communicator of the slab exchange could not be built on every rank%s" % (": %s" % created if created else ""))

    def all_gather(self, send, recv, stream):
        if not (send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous()):
            raise TypeError("SlabExchange works on contiguous CUDA tensors")
        if recv.numel() != send.numel() * self.world or recv.dtype != send.dtype:
            raise ValueError("recv must hold world x send elements of the same dtype")
        nbytes = send.numel() * send.element_size()
        if nbytes % 8:
            raise ValueError("SlabExchange moves 8-byte records")
        C = self._C
        self._check(self._lib.tsdf_slab_exchange_all_gather(self._h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), nbytes // 8,
                                                            C.c_void_p(int(stream))))

    def ranks_seen(self):
        """How many ranks the communicator itself reports (ncclCommCount; the world of a caller's own collective)."""
        n = self._C.c_int()
        self._check(self._lib.tsdf_slab_exchange_ranks_seen(self._h, self._C.byref(n)))
        return n.value

    def validate_merge(self, slab_volume, width, height, cam, merged_vertices, merged_normals=None):
        """SURVEY.md 8e mode B (tsdf_slab_validate_merge): all-gather of the ranks' distance slabs, the whole volume's ordinary ray cast
        on every rank, and the number of words in which it differs from the merged picture (CUDA tensors).  Collective; 0 = equal."""
        import numpy as np
        C = self._C
        pose = np.ascontiguousarray(cam.pose(), np.float32).reshape(-1)
        kinv = np.ascontiguousarray(cam.kinv(), np.float32).reshape(-1)
        fp = C.POINTER(C.c_float)
        n = C.c_uint64()
        self._check(self._lib.tsdf_slab_validate_merge(slab_volume._h, self._h, width, height, pose.ctypes.data_as(fp), kinv.ctypes.data_as(fp),
                                                       C.c_void_p(merged_vertices.data_ptr()),
                                                       C.c_void_p(merged_normals.data_ptr()) if merged_normals is not None else None, C.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tsdf_slab_exchange_destroy(self._h)
            self._h = self._C.c_void_p()


class LoopbackExchange(SlabExchange):
    """"Rank `rank` of `world`" on a box with one GPU (tsdf_slab_exchange_create_loopback): the all-gather copies this rank's own records
    into every rank's place on the caller's stream.  For emulating one rank's step of a P-GPU run (tools/dbg_slab_pipeline.py) and for
    tests; the merged picture holds this slab's hits only."""

    def __init__(self, rank, world):
        import ctypes as C
        from ._capi import check, lib
        self._C, self._lib, self._check = C, lib, check
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        self._cb = None
        check(lib.tsdf_slab_exchange_create_loopback(self.rank, self.world, C.byref(self._h)))


StreamAllGather = SlabExchange      # (round-2 name)
