"""world_size-2 gloo tests (CPU) of the data-parallel host logic of the training step: the ian_comm_ops callback table
(trainer.build_ops <- trainer.Comm.ops) driven the way the C sequencer drives it (csrc/ian_trainer.cpp: fire / finish_allreduce /
allreduce_ordered / gather) -- raw buffer addresses through the C function pointers, here on HOST buffers -- the all-rank
fallback agreement of NativeRcclComm, and bench.py's own multi-rank launch."""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tree(parts):
    """the pairwise tree of ian_k_tree_sum (kernels_train.hip tree_sum_seq): T(lo,hi) = T(lo,lo+m) + T(lo+m,hi), m = the largest
    power of two below hi-lo"""
    if len(parts) == 1:
        return parts[0]
    m = 1
    while m * 2 < len(parts):
        m *= 2
    return _tree(parts[:m]) + _tree(parts[m:])


def _drive_table(comm, ops, rank, world):
    """What one data-parallel update asks of the table, with the sequencer's call pattern and host buffers."""
    rs = np.random.RandomState(0)
    # (1) gradient buckets: a flat group cut into bucket_bytes pieces, each handed to allreduce_sum as it becomes ready (here: in
    # reverse order, as backward produces them), then ONE wait_all before the optimiser reads the gradients
    full = rs.randn(world, 1777).astype(np.float32)
    g = full[rank].copy()
    step = comm.bucket_bytes // 4
    cuts = [(o, min(g.size, o + step)) for o in range(0, g.size, step)]
    assert len(cuts) >= 7
    for lo, hi in reversed(cuts):
        assert ops.allreduce_sum(ops.ctx, g.ctypes.data + 4 * lo, hi - lo, None) == 0
    assert ops.wait_all(ops.ctx, None) == 0
    assert ops.wait_all(ops.ctx, None) == 0                      # nothing pending: a no-op, not an error
    ok1 = np.allclose(g, full.sum(0), rtol=1e-6, atol=1e-6)
    # (2) SyncBN: float64 [2][C] sums travel as pairs of floats (allreduce_ordered: count = 2 * width), every rank combines the
    # gathered rows in RANK ORDER with the pairwise tree -> all ranks hold bitwise the same statistics
    x = rs.randn(world * 8, 16).astype(np.float32)
    local = x[rank * 8:(rank + 1) * 8].astype(np.float64)
    sums = np.concatenate([local.sum(0), (local ** 2).sum(0)])
    gbuf = np.zeros((world, sums.size), np.float64)
    assert ops.allgather(ops.ctx, sums.ctypes.data, gbuf.ctypes.data, 2 * sums.size, None) == 0
    comb = _tree([gbuf[r] for r in range(world)])
    want = _tree([np.concatenate([x[r * 8:(r + 1) * 8].astype(np.float64).sum(0), (x[r * 8:(r + 1) * 8].astype(np.float64) ** 2).sum(0)])
                  for r in range(world)])
    ok2 = np.array_equal(comb, want)
    # (3) MinibatchLayer: rank r owns rows [r*n, (r+1)*n) of the global activation matrix
    act = x[rank * 8:(rank + 1) * 8].copy()
    act_all = np.zeros_like(x)
    assert ops.allgather(ops.ctx, act.ctypes.data, act_all.ctypes.data, act.size, None) == 0
    ok3 = np.array_equal(act_all, x)
    # (4) a failing callback must come back as an error CODE (the step then fails with -30), never as an exception across C
    bad = ops.allreduce_sum(ops.ctx, 0, 16, None)
    ok4 = bad != 0 and len(comm.errors) == 1
    return ok1, ok2, ok3, ok4


def _worker(rank, world, port, q, native):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neural_photo_editor_amd import trainer as T
        if native:
            # the N > 1 default of bench.py / train_cli.py: the librccl filler; no GPU here (host buffers), so every rank must agree
            # to fall back to the torch.distributed filler -- together
            comm = T.NativeRcclComm(bucket_bytes=1000)
            ops = comm.ops(torch, host=True)
            assert comm.filler.startswith("torch.distributed (fallback:"), comm.filler
            assert comm.gather_group is not None
        else:
            comm = T.Comm(bucket_bytes=1000)                       # tiny buckets: 250 floats -> 8 pieces
            assert comm.gather_group is not None and comm.gather_group is not comm.group     # the all-gathers' own group
            ops = comm.ops(torch, host=True)
        assert (ops.world, ops.rank) == (world, rank) == (comm.world, comm.rank)
        q.put((rank,) + _drive_table(comm, ops, rank, world))
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("native", [False, True])
def test_comm_table_world2_gloo(native):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, native)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(all(r[1:]) for r in res), res


def test_comm_single_process():
    """No process group: world 1, and the all-gather of the table is a copy."""
    from neural_photo_editor_amd import trainer as T
    c = T.Comm()
    assert c.world == 1 and c.rank == 0 and c.filler == "torch.distributed"
    t = torch.arange(5, dtype=torch.float32)
    out = torch.zeros(5)
    c.all_gather_rows(t, out)
    assert torch.equal(out, t)
    assert isinstance(T.default_comm(), T.Comm) and not isinstance(T.default_comm(), T.NativeRcclComm)


def test_ian_comm_ops_layout_matches_the_header():
    """trainer.CommOps is the ctypes mirror of ian_comm_ops (include/ian_train.h): two int32, a context pointer, three function
    pointers, in that order."""
    from neural_photo_editor_amd import trainer as T
    names = [f[0] for f in T.CommOps._fields_]
    assert names == ["world", "rank", "ctx", "allreduce_sum", "wait_all", "allgather"]
    assert C.sizeof(T.CommOps) == 8 + 4 * C.sizeof(C.c_void_p)
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "ian_train.h")).read()
    body = hdr[hdr.index("typedef struct ian_comm_ops {"):hdr.index("} ian_comm_ops;")]
    order = [body.index(k) for k in ("world, rank", "void* ctx", "(*allreduce_sum)", "(*wait_all)", "(*allgather)")]
    assert order == sorted(order)


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (torch.distributed.run on
    127.0.0.1), rendezvous, take the max over ranks and print ONE line with n_gpus == 2.  --dry-run keeps the GPU out
    of it so the control flow is covered on CPU (gloo); the timed HIP legs are the same code after the rendezvous."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env["IAN_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 3 and out["dry_run"] is True


def _run_bench_dry(extra_env, timeout_s):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env["IAN_BENCH_BACKEND"] = "gloo"
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--train", "--train-timeout", str(timeout_s),
                        "--steps", "2", "--warmup", "1"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_train_leg_ladder_first_mode():
    """N > 1 training leg of bench.py (round-5 verdict item 4): every attempt runs in child processes with their own rendezvous.  Healthy
    case: the first collective mode (librccl, two communicators) answers, one attempt, the line says which mode ran."""
    out = _run_bench_dry({}, 120)
    t = out["train_step"]
    assert t["comm_mode"] == "native2" and t["ranks_seen"] == 2 and t["dry_run"] is True
    assert [a["mode"] for a in t["attempts"]] == ["native2"] and t["attempts"][0]["ok_on_all_ranks"] is True


def test_train_leg_ladder_retries_after_a_hang():
    """A rank that never arrives in a collective (IAN_BENCH_FAKE_HANG: the last rank of the named modes sleeps) must cost ONE attempt, not
    the run: the children of that attempt are killed at the per-attempt timeout, all parents agree, the next mode is tried -- here the
    two-communicator mode and the one-communicator mode hang, the torch.distributed filler answers -- and the headline line is intact."""
    out = _run_bench_dry({"IAN_BENCH_FAKE_HANG": "native2,native1"}, 12)
    t = out["train_step"]
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert t["comm_mode"] == "torch" and t["ranks_seen"] == 2
    assert [a["mode"] for a in t["attempts"]] == ["native2", "native1", "torch"]
    assert [a["ok_on_all_ranks"] for a in t["attempts"]] == [False, False, True]
    assert "timeout" in t["attempts"][0]["outcome_rank0"] and "timeout" in t["attempts"][1]["outcome_rank0"]


def test_train_leg_ladder_gives_up_with_the_headline_intact():
    out = _run_bench_dry({"IAN_BENCH_FAKE_HANG": "native2,native1,torch"}, 8)
    t = out["train_step"]
    assert out["n_gpus"] == 2 and "error" in t and len(t["attempts"]) == 3 and not any(a["ok_on_all_ranks"] for a in t["attempts"])


def _stage_worker(rank, world, port, q, fail_spec, one_comm):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["IAN_COMM_TEST_FAIL"] = fail_spec
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from neural_photo_editor_amd import trainer as T
        comm = T.NativeRcclComm(bucket_bytes=1000, one_comm=one_comm)
        comm.ops(torch, host=False)        # not driven: only the set-up agreement is under test (no GPU here)
        q.put((rank, comm.filler, list(comm.stages), comm.gather_group is comm.group or comm.gather_group is None))
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("one_comm", [False, True])
def test_native_comm_setup_is_staged_and_all_ranks_fall_back_together(one_comm):
    """a rank other than 0 that cannot use librccl must be found out BEFORE anybody enters the blocking
    ncclCommInitRank.  Rank 1's stage-0 probe is made to fail (IAN_COMM_TEST_FAIL=available@1); both ranks must leave through the
    torch.distributed filler after the FIRST agreement, with the reason naming rank 1 -- and in one-communicator mode the fallback
    must not create a second process group either."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stage_worker, args=(r, world, port, q, "available@1", one_comm)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, filler, stages, shared_group in res:
        assert filler.startswith("torch.distributed (fallback: rank 1:"), filler
        assert stages == [("available", False)], stages
        assert shared_group == one_comm


def test_default_comm_modes():
    """trainer.default_comm: mode names are validated, and without an RCCL process group every mode is the torch.distributed filler;
    IAN_RCCL_ONE_COMM=1 removes the second process group of the torch filler as well."""
    from neural_photo_editor_amd import trainer as T
    assert T.COMM_MODES == ("native2", "native1", "torch")
    for m in T.COMM_MODES:
        assert type(T.default_comm(mode=m)) is T.Comm
    with pytest.raises(ValueError):
        T.default_comm(mode="nccl3")
    os.environ["IAN_RCCL_ONE_COMM"] = "1"
    try:
        c = T.default_comm()
        assert c.gather_group is None and "one process group" in c.filler
        assert T.NativeRcclComm().one_comm is True
    finally:
        del os.environ["IAN_RCCL_ONE_COMM"]
    assert T.NativeRcclComm().one_comm is False and T.NativeRcclComm(one_comm=True).filler == "librccl (native, 1 communicator)"


def test_local_rccl_abi_declarations_match_the_installed_header():
    """csrc/ian_comm_rccl.cpp declares the slice of the NCCL ABI it calls itself (no rccl.h at build time) and
    resolves the entry points with dlopen; with more than one rank those declarations run for the first time on the 8-GPU node, so
    they are pinned here against the header of the RCCL this image ships: id size, result / datatype / reduction codes and the
    argument lists of the five entry points."""
    import re
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    h = open(hdr).read()
    src = open(os.path.join(ROOT, "neural_photo_editor_amd", "csrc", "ian_comm_rccl.cpp")).read()
    const = lambda name: int(re.search(r"\b%s\s*=\s*(\d+)" % name, src).group(1))
    assert int(re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+(\d+)", h).group(1)) == const("kIdBytes") == 128
    assert re.search(r"typedef struct \{ char internal\[NCCL_UNIQUE_ID_BYTES\];", h)          # passed BY VALUE to ncclCommInitRank
    for ours, theirs in (("kRcclSuccess", "ncclSuccess"), ("kRcclFloat32", "ncclFloat32"), ("kRcclSum", "ncclSum")):
        assert int(re.search(r"\b%s\s*=\s*(\d+)" % theirs, h).group(1)) == const(ours), theirs

    def header_args(fn):
        m = re.search(r"ncclResult_t\s+%s\(([^;]*?)\);" % fn, h, re.S)
        kinds = []
        for a in m.group(1).split(","):
            t = " ".join(a.split()[:-1]).replace("const ", "")            # drop the parameter name
            star = "*" in a
            kinds.append({"void": "ptr", "ncclUniqueId": "id*" if star else "id", "ncclComm_t": "comm*" if star else "comm", "int": "int",
                          "size_t": "size", "ncclDataType_t": "int", "ncclRedOp_t": "int", "hipStream_t": "stream"}[t.replace("*", "").strip()])
        return kinds

    def our_args(member):
        m = re.search(r"\(\*%s\)\(([^)]*)\)" % member, src)
        return [{"const void*": "ptr", "void*": "ptr", "RcclUniqueId*": "id*", "RcclUniqueId": "id", "RcclComm*": "comm*", "RcclComm": "comm",
                 "int": "int", "size_t": "size", "hipStream_t": "stream"}[a.strip()] for a in m.group(1).split(",")]

    for member, fn in (("GetUniqueId", "ncclGetUniqueId"), ("CommInitRank", "ncclCommInitRank"), ("CommDestroy", "ncclCommDestroy"),
                       ("AllReduce", "ncclAllReduce"), ("AllGather", "ncclAllGather")):
        assert our_args(member) == header_args(fn), (fn, our_args(member), header_args(fn))
        assert '"%s"' % fn in src                                                              # the name handed to dlsym
