"""The C-ABI library loads on a CPU-only box and exports every symbol include/mocca.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mocca.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mocca_[a-z_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def built():
    from mocca_envs_amd.build import build_lib
    return build_lib()


def test_header_and_binding_agree(built):
    from mocca_envs_amd import lib
    assert _declared() == sorted(lib.SYMBOLS), "include/mocca.h and mocca_envs_amd/lib.py list different entry points"


def test_library_exports_every_symbol(built):
    so = C.CDLL(built)
    for name in _declared():
        assert hasattr(so, name), f"{name} missing from libmocca_hip.so"


def test_load_checks_version_and_model_layout(built):
    from mocca_envs_amd import lib, model as M
    l = lib.load()
    assert l.mocca_abi_version() == lib.ABI_VERSION
    assert l.mocca_model_sizeof() == C.sizeof(M.MoccaModel)  # ctypes mirror == struct MoccaModel


def test_bad_arguments_are_errors_not_crashes(built):
    from mocca_envs_amd import lib, model as M
    l = lib.load()
    h = C.c_void_p()
    blob = M.compile_walker3d().to_bytes()
    assert l.mocca_create(blob, len(blob) - 1, 0, 4, 0, C.byref(h)) == -1      # wrong size
    assert l.mocca_create(blob, len(blob), 7, 4, 0, C.byref(h)) == -1          # unknown task
    bad = bytearray(blob); bad[0] ^= 0xFF
    assert l.mocca_create(bytes(bad), len(bad), 0, 4, 0, C.byref(h)) == -1     # bad magic
    m = M.compile_walker3d(); m.parent[5] = 2
    assert l.mocca_create(m.to_bytes(), len(blob), 0, 4, 0, C.byref(h)) == -3  # topology mismatch
    assert l.mocca_last_error(None)
    assert l.mocca_step(None, None, None, None, None, None, None) == -1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mocca_envs_amd import lib
    from mocca_envs_amd.vec_env import VecEnv
    with pytest.raises(lib.MoccaError):
        VecEnv("Walker3DCustomEnv-v0", 4)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under mocca_envs_amd/ may import, include, link or dlopen it."""
    pkg = os.path.join(ROOT, "mocca_envs_amd")
    bad = re.compile(r"^\s*(import\s+oracle|from\s+oracle\b)|#include\s*[<\"][^>\"]*oracle|liboracle|mocca_oracle\.c\b(?!\s+so)", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f)).read()
                m = bad.search(txt)
                assert m is None, f"{f}: {m.group(0)!r}"


def test_diagnostic_builds_are_refused(tmp_path):
    """A library compiled with a profiling switch (-DMOCCA_SKIP_* skips a physics phase, MOCCA_STAMPS adds timing stores) says so
    through mocca_is_diagnostic_build(), and the binding refuses it unless MOCCA_ALLOW_DIAGNOSTIC_BUILD is set: a mis-set -D
    cannot ship a kernel that is wrong by construction."""
    import subprocess
    import sys
    from mocca_envs_amd.build import build_lib
    so = build_lib(force=True, extra_flags=["-DMOCCA_SKIP_SOLVE"], out=str(tmp_path / "libdiag.so"))
    assert C.CDLL(so).mocca_is_diagnostic_build() == 1
    code = "from mocca_envs_amd import lib; lib.load()"
    env = dict(os.environ, MOCCA_LIB_PATH=so, PYTHONPATH=ROOT)
    env.pop("MOCCA_ALLOW_DIAGNOSTIC_BUILD", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "diagnostic build" in r.stderr
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, MOCCA_ALLOW_DIAGNOSTIC_BUILD="1"), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    from mocca_envs_amd import lib
    assert lib.load().mocca_is_diagnostic_build() == 0          # the product build is not one


def test_multi_handle_envs_validate_their_arguments_before_touching_a_gpu():
    """mocca_envs_amd.multi: the cuts must be even, and sub-batches do not combine with several devices (shard first)."""
    import pytest
    from mocca_envs_amd.multi import ShardedVecEnv, SubBatchedVecEnv, make_vec_env
    with pytest.raises(ValueError):
        SubBatchedVecEnv("Walker3DCustomEnv-v0", 100, sub_batches=3)
    with pytest.raises(ValueError):
        ShardedVecEnv("Walker3DCustomEnv-v0", 100, devices=[0, 1, 2])
    with pytest.raises(ValueError):
        make_vec_env("Walker3DCustomEnv-v0", 128, sub_batches=2, devices=[0, 1])


def test_trainer_lazy_done_and_infos_read_the_record_ring():
    """trainer_api._LazyDone / _Infos over one slot of the episode-record ring (include/mocca.h mocca_episode_rec): what `for d in done` /
    `for info in infos` / `infos[i]` / `len(infos)` of the PPO trainers see.  Nothing is read before the first look; records whose serial is
    another step's (left over in the slot) are not this step's."""
    import numpy as np
    from mocca_envs_amd.trainer_api import _Infos, _LazyDone, _StepRecords

    class Ev:
        waits = 0

        def synchronize(self):
            Ev.waits += 1

    class Env:
        num_envs, _slots, _want_terminal, _stepper = 12, 4, False, True
        _events = [Ev() for _ in range(4)]
        _ring = np.zeros((4, 12, 4), np.int32)

    def put(slot, env, serial, ret, length, flags):
        Env._ring[slot, env] = (serial, np.float32(ret).view(np.int32), length, flags)

    serial = 6                                   # slot 6 % 4 = 2
    put(2, 3, serial, 1.5, 7, 1 | (4 << 8))      # terminated, steps_reached 4
    put(2, 9, serial, -2.0, 1000, 3 | (19 << 8)) # terminated AND at the TimeLimit: bad_transition (a2c-ppo-acktr's TimeLimitMask)
    put(2, 5, serial - 4, 9.0, 3, 1)             # left over from four steps ago
    rec = _StepRecords(Env, serial)
    done, infos = _LazyDone(rec), _Infos(12, rec)
    assert len(done) == 12 and len(infos) == 12 and Ev.waits == 0       # nothing fetched yet
    fin = dict(infos.finished())
    assert Ev.waits == 1 and sorted(fin) == [3, 9]
    assert fin[3] == {"episode": {"r": 1.5, "l": 7}, "steps_reached": 4}
    assert fin[9] == {"episode": {"r": -2.0, "l": 1000}, "steps_reached": 19, "bad_transition": True, "TimeLimit.truncated": True}
    assert infos[3]["episode"]["l"] == 7 and infos[0] == {} and infos[-3] is fin[9]
    assert [("episode" in i) for i in infos] == [k in fin for k in range(12)]
    assert sum("bad_transition" in i.keys() for i in infos) == 1
    assert [len(x) for x in infos[2:5]] == [0, 2, 0]
    d = np.asarray(done)
    assert d.dtype == bool and d.tolist() == [k in fin for k in range(12)] and done.sum() == 2 and bool(done[9]) and not done[5]
    assert [[0.0] if x else [1.0] for x in done] == [[0.0] if k in fin else [1.0] for k in range(12)]     # the trainers' mask comprehension
    assert (~done).sum() == 10 and (done == d).all() and Ev.waits == 1
    Env._ring[2] = 0                             # the slot is rewritten later: the fetched step keeps its records
    assert dict(infos.finished()) == fin
    import pytest
    with pytest.raises(IndexError):
        infos[12]
