"""Multi-GPU sharding: planner + hit-record exchange, exercised with world_size-2/3 gloo process groups on CPU.
The per-rank compute in these CPU tests is the oracle's slab ray caster (the checker); what is under test is
the product's host logic in tsdf_amd/multi.py (slab planning, the all-gather) and the protocol itself: merged
slab records must reproduce the single-volume ray cast bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_ranges_partition_the_grid():
    from tsdf_amd.multi import owner_of_plane, resident_range, slab_range
    for Z in (1, 7, 8, 64, 512, 1000):
        for world in (1, 2, 3, 4, 8):
            if world > Z:
                with pytest.raises(ValueError):
                    slab_range(Z, world, 0)
                continue
            prev = 0
            sizes = []
            for r in range(world):
                zb, ze = slab_range(Z, world, r)
                assert zb == prev and ze > zb
                prev = ze
                sizes.append(ze - zb)
                lo, hi = resident_range(Z, world, r)
                assert lo == zb and hi == min(ze + 1, Z)
                assert owner_of_plane(Z, world, zb) == r and owner_of_plane(Z, world, ze - 1) == r
            assert prev == Z and max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        slab_range(8, 2, 2)


def test_balanced_and_refined_slab_ranges():
    """The work-balanced planner: contiguous, complete, at least min_planes each, never worse than the uniform split under its
    own cost model, and the measured refinement moves planes from the slow ranks to the fast ones."""
    from tsdf_amd.multi import balanced_slab_ranges, refine_slab_ranges, slab_range
    rng = np.random.default_rng(3)
    for Z, world in ((512, 8), (512, 4), (1024, 8), (100, 3), (64, 8), (9, 2)):
        cost = rng.uniform(0.0, 0.05, Z)
        cost[Z // 3: Z // 2] += 1.0                                    # the surfaces sit in a band of planes
        for min_planes in (1, 8):
            if world * min_planes > Z:
                with pytest.raises(ValueError):
                    balanced_slab_ranges(cost, world, min_planes)
                continue
            r = balanced_slab_ranges(cost, world, min_planes)
            assert len(r) == world and r[0][0] == 0 and r[-1][1] == Z
            assert all(a[1] == b[0] for a, b in zip(r, r[1:])) and all(e - b >= min_planes for b, e in r)
            worst = max(cost[b:e].sum() for b, e in r)
            uniform = max(cost[slice(*slab_range(Z, world, k))].sum() for k in range(world))
            assert worst <= uniform + 1e-9
    assert balanced_slab_ranges(np.zeros(16), 4) == [slab_range(16, 4, k) for k in range(4)]      # nothing known: the uniform split
    # measured refinement: rank 2 was three times slower than the others -> its slab shrinks, the plan stays a partition
    ranges = [slab_range(512, 4, k) for k in range(4)]
    new = refine_slab_ranges(ranges, [0.10, 0.10, 0.30, 0.10], 512, min_planes=8)
    assert new[0][0] == 0 and new[-1][1] == 512 and all(a[1] == b[0] for a, b in zip(new, new[1:]))
    assert new[2][1] - new[2][0] < 128 and all(e - b >= 8 for b, e in new)
    # every rank computes the same plan from the same numbers
    assert new == refine_slab_ranges(list(ranges), [0.10, 0.10, 0.30, 0.10], 512, min_planes=8)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, W, H, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle as O
        from tsdf_amd.multi import gather_hits, resident_range, slab_range
        k, kinv = O.camera_k(591.1 / 4, 590.1 / 4, 331.0 / 4, 234.6 / 4)
        frames = []
        for i in range(3):
            pose = O.look_at(O.identity_pose((1100 + 250 * i, 1600 - 150 * i, -800)), (1500, 1500, 1600))
            yy, xx = np.mgrid[0:H, 0:W]
            depth = (2300 + 200 * np.sin(xx / 23.0 + i) + 100 * np.cos(yy / 9.0)).astype(np.uint16).reshape(-1)
            frames.append((depth, pose))
        zb, ze = slab_range(n, world, rank)
        lo, hi = resident_range(n, world, rank)
        slab = O.Volume((n, n, n), (3000, 3000, 3000), z_store=(lo, hi))
        for depth, pose in frames:          # each rank integrates its planes + halo, no communication
            slab.integrate(depth, W, H, O.mat4_inverse(pose), k, kinv)
        pose = frames[0][1]
        mine = torch.from_numpy(slab.raycast_slab(W, H, pose, kinv, (zb, ze)).view(np.float32))   # {k, t} records as words
        allh = gather_hits(mine).numpy().view(np.uint32)           # (world, W*H, 2) on every rank
        V = slab.merge_hits(allh, W, H, pose, kinv)               # per-pixel min-k select, vertex from t and the pixel's ray
        if rank == 0:
            whole = O.Volume((n, n, n), (3000, 3000, 3000))
            for depth, p in frames:
                whole.integrate(depth, W, H, O.mat4_inverse(p), k, kinv)
            Vw, Nw = whole.raycast(W, H, pose, kinv)
            same = (V.view(np.uint32) == Vw.view(np.uint32)) | (np.isnan(V) & np.isnan(Vw))
            ks = allh[:, :, 0].astype(np.int64)
            finite = ks != 0xffffffff
            np.save(os.path.join(tmp, "result.npy"),
                    np.array([int(same.all()), int((~np.isnan(Vw[:, 0])).sum()),
                              # a sample has exactly one owner: no two ranks may report the same finite k
                              int(((ks == ks.min(axis=0)) & finite).sum(axis=0).max())]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_records_gathered_over_gloo_reproduce_the_single_volume_raycast(tmp_path, world, oracle):
    n, W, H = 48, 160, 120
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, W, H, str(tmp_path)), nprocs=world, join=True)
    ok, hits, max_owners = np.load(os.path.join(str(tmp_path), "result.npy"))
    assert ok == 1
    assert hits > 1000
    assert max_owners == 1


def test_merge_hits_refuses_cpu_tensors():
    from tsdf_amd.multi import merge_hits
    with pytest.raises(TypeError):
        merge_hits(None, torch.zeros((2, 4, 2)), 2, 2, None)


def _mesh_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import tsdf_amd
        from tsdf_amd.multi import gather_vertices, slab_range
        size, vs = (20, 12, 23), (10.0, 12.5, 9.0)
        D = np.random.default_rng(11).uniform(-1.0, 1.0, size[0] * size[1] * size[2]).astype(np.float32)
        whole = tsdf_amd.marching_cubes(D, size, vs)                         # host marching cubes, the whole grid
        # this rank's part: the cube layers rooted in its planes = the host marching cubes of planes [zb, ze] (with the
        # halo plane), shifted to its place -- what TSDFVolume.extract_surface returns for the slab on a GPU
        zb, ze = slab_range(size[2], world, rank)
        hi = min(ze + 1, size[2])
        planes = D.reshape(size[2], -1)[zb:hi]
        mine = tsdf_amd.marching_cubes(planes.ravel(), (size[0], size[1], hi - zb), vs, offset=(0.0, 0.0, 0.0)) if hi - zb >= 2 else np.zeros((0, 3), np.float32)
        mine = mine.copy()
        # (z of a slab-local vertex: the same expression with the global plane index -- recompute rather than add an offset)
        got = gather_vertices(torch.from_numpy(mine)).numpy()
        if rank == 0:
            np.save(os.path.join(tmp, "mesh.npy"), np.array([got.shape[0], whole.shape[0],
                                                             int(np.array_equal(got[:, :2].view(np.uint32), whole[:, :2].view(np.uint32)))]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_slab_meshes_gathered_over_gloo_are_the_whole_mesh_in_cube_order(tmp_path, world):
    """gather_vertices: parts of different lengths, concatenated in rank order (x and y of every vertex compared; z is
    produced with the global plane index by the device kernel, tests/test_parity_marching_cubes.py)."""
    port = _free_port()
    mp.spawn(_mesh_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n_got, n_whole, same_xy = np.load(os.path.join(str(tmp_path), "mesh.npy"))
    assert n_got == n_whole and n_whole > 100 and same_xy == 1


@pytest.mark.gpu
def test_mode_b_validator_counts_the_words_that_differ(oracle):
    """tsdf_slab_validate_merge (SURVEY.md 8e mode B) in one process: a slab that is the whole grid, an exchange of one rank over a
    caller's collective.  The merged picture of the slab path passes with 0 differing words; the same picture with three words
    changed is counted as 3; a NaN (no hit) only equals a NaN."""
    import ctypes as C
    import tsdf_amd
    from tsdf_amd import _capi, synth
    from tsdf_amd.multi import device_words
    from tests.helpers import H, W
    lib, check = _capi.lib, _capi.check
    n = 64
    vol = tsdf_amd.TSDFVolume((n, n, n), (3000.0,) * 3, slab=(0, n))
    bil = tsdf_amd.BilateralFilter(30.0, 4.5)
    cam = None
    for i in range(3):
        depth, cam = synth.depth_frame(i, 30, seed=0x5EED0004)
        f = depth.copy()
        bil.filter(f, W, H)
        vol.integrate(f, W, H, cam)

    def gather(user, mine, allr, n_pixels, stream):     # world of one: rank 0's records are everybody's
        device_words(allr, 2 * n_pixels).copy_(device_words(mine, 2 * n_pixels))
        torch.cuda.synchronize()
        return 0
    cb = _capi.EXCHANGE_FN(gather)
    h = C.c_void_p()
    check(lib.tsdf_slab_exchange_create_callback(0, 1, cb, None, C.byref(h)))
    seen = C.c_int()
    check(lib.tsdf_slab_exchange_ranks_seen(h, C.byref(seen)))
    assert seen.value == 1
    rc = tsdf_amd.GPURaycaster(W, H)
    hits = torch.empty((H * W, 2), dtype=torch.int32, device="cuda")
    V = torch.empty((H * W, 3), dtype=torch.float32, device="cuda")
    N = torch.empty_like(V)
    rc.raycast_slab_device(vol, cam, hits.data_ptr())
    tsdf_amd.merge_hits_normals_device(vol, hits.data_ptr(), 1, W, H, cam, V.data_ptr(), N.data_ptr(), 0)
    torch.cuda.synchronize()
    assert int((~torch.isnan(V[:, 0])).sum().item()) > 1000 and int(torch.isnan(V[:, 0]).sum().item()) > 0

    def differing(v, nrm):
        fp = C.POINTER(C.c_float)
        out = C.c_uint64()
        pose, kinv = np.ascontiguousarray(cam.pose(), np.float32), np.ascontiguousarray(cam.kinv(), np.float32)
        check(lib.tsdf_slab_validate_merge(vol._h, h, W, H, pose.ctypes.data_as(fp), kinv.ctypes.data_as(fp), C.c_void_p(v.data_ptr()),
                                           C.c_void_p(nrm.data_ptr()) if nrm is not None else None, C.byref(out)))
        return int(out.value)
    assert differing(V, N) == 0
    assert differing(V, None) == 0
    bad = V.clone()
    hit = int(torch.nonzero(~torch.isnan(V[:, 0]))[0].item()); miss = int(torch.nonzero(torch.isnan(V[:, 0]))[0].item())
    bad[hit, 0] += 1.0                       # a vertex off by a millimetre
    bad[hit + 1, 2] = float("nan")           # a hit turned into a miss (or a miss stays one)
    bad[miss, 1] = 0.0                       # a miss turned into a number
    expected = 2 + (0 if bool(torch.isnan(V[hit + 1, 2]).item()) else 1)
    assert differing(bad, N) == expected
    lib.tsdf_slab_exchange_destroy(h)
    vol.close()
