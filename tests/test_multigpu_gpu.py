"""GPU, two or more devices: the RCCL path of SURVEY.md section 8(e) EXECUTED -- dfvo_comm_create + dfvo_allgather_poses (the
C ABI's communicator: ncclAllGather of padded [nmax, 17] f64 rows, per-rank trim) with uneven and empty per-rank counts against
dist.allgather_rows over gloo on the same rows, and `bench.py --gpus 2` through torch.distributed.run (one rank per device,
backend nccl = RCCL) asserting that both ranks took part and that the gathered trajectory equals the single-rank re-track.
Skips cleanly on a one-device box (every box this repository has been run on so far: the driver's SCALE / MULTICHIP records
are 'skipped: no 8-GPU node').  No reference counterpart: DF-VO is single-process (libs/dfvo.py:109-119,157-161 is the
recurrence the gathered rows feed)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, json, os, sys
import numpy as np
import torch
import torch.distributed as td
sys.path.insert(0, %(root)r)
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(int(os.environ["LOCAL_RANK"]))
td.init_process_group("gloo", rank=rank, world_size=world)   # out-of-band channel + the reference gather
D = importlib.import_module("df-vo_amd.dist")
comm = D.RcclComm.from_torch(td, world, rank)
ok = True
for case, counts in enumerate(json.loads(os.environ["DFVO_TEST_COUNTS"])):
    rng = np.random.RandomState(100 * case + rank)
    rows = rng.randn(counts[rank], 17)
    rows[:, 16] = rng.randint(0, 2, counts[rank])
    got = comm.allgather_rows(rows, counts)                   # dfvo_allgather_poses: RCCL over xGMI / PCIe
    want = D.allgather_rows(rows, counts, world, rank, dist=td)  # gloo, host tensors
    same = got.shape == want.shape and np.array_equal(got, want)
    off = int(sum(counts[:rank]))
    mine = np.array_equal(got[off:off + counts[rank]], rows)
    ok = ok and same and mine
    print("rank", rank, "counts", counts, "equal to gloo:", same, "own rows in place:", mine, flush=True)
comm.close()
flag = torch.tensor([1 if ok else 0])
td.all_reduce(flag, op=td.ReduceOp.MIN)
td.destroy_process_group()
sys.exit(0 if int(flag.item()) == 1 else 1)
'''


def _n_devices():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _launch(nproc, argv, env_extra, port, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + argv
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 3])
def test_rccl_allgather_of_uneven_pose_rows_equals_gloo(gpu, tmp_path, world):
    if _n_devices() < world:
        pytest.skip("needs %d visible GPUs (one RCCL rank per device); this box has %d" % (world, _n_devices()))
    cases = [[3, 2], [4, 0], [0, 5], [1, 1]] if world == 2 else [[3, 2, 4], [4, 0, 1], [0, 0, 2]]
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER % {"root": ROOT})
    r = _launch(world, [str(script)], {"DFVO_TEST_COUNTS": json.dumps(cases)}, 29540 + world)
    print(r.stdout[-2000:])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("equal to gloo: True") == world * len(cases)


def test_bench_two_gpus_over_rccl(gpu):
    if _n_devices() < 2:
        pytest.skip("needs 2 visible GPUs; this box has %d" % _n_devices())
    r = _launch(2, ["bench.py", "--gpus", "2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-exact-leg", "--no-other-legs"],
                {}, 29547)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print({k: line[k] for k in ("value", "n_gpus", "ms_per_step", "scaling")}, line.get("sequence_check"), line["config"].get("ranks_seen"))
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["ranks_seen"] == 2
    assert line["sequence_check"]["equal_to_single_rank_run"] is True
