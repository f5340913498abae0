"""Extended parity fuzz beyond the seeds tests/test_fuzz_parity.py pins (GPU box): python tools/extra_fuzz.py"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import oracle as orc
orc.build()
import tests.test_fuzz_parity as t
fails = 0
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
hi = int(sys.argv[2]) if len(sys.argv) > 2 else lo + 80
for seed in range(lo, hi):
    for fn in (t.test_random_distance_fields, t.test_smooth_distance_fields, t.test_random_scene, t.test_axis_aligned_cameras_on_integer_principal_points):
        try:
            fn(orc, seed)
        except AssertionError as e:
            fails += 1
            print("FAIL", fn.__name__, seed, str(e)[:200])
        except Exception as e:
            print("ERR", fn.__name__, seed, repr(e)[:200])
    try:
        t.test_random_slab_splits_equal_the_whole_volume(seed)
    except AssertionError as e:
        fails += 1
        print("FAIL slab splits", seed, str(e)[:200])
print("extra fuzz done, failures:", fails)
