"""CPU tests of the host side: config recorder, lowering, MADE masks (product code), checkpoint format,
and that the C-ABI library builds, loads and exports every symbol include/ian.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from neural_photo_editor_amd import checkpoints, config_loader as cl, lowering, made
from neural_photo_editor_amd import lib as L
from oracle import ian_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "neural_photo_editor_amd", "configs")
REF = "/root/reference"


def lowered(path, dnn=True):
    mod = cl.load_config(path)
    return mod, lowering.lower_model(cl.build_model(mod, dnn=dnn))


def op_signature(low):
    return [(o.kind, o.segment, o.src, o.src2, o.src3, o.dst, o.cin, o.cout, o.in_h, o.in_w, o.act, o.has_bias,
             tuple(o.flat), tuple(o.unflat), tuple(o.scales), o.name, o.bn_name) for o in low.ops]


def test_ian_simple_lowering():
    mod, low = lowered(os.path.join(CFG, "IAN_simple.py"))
    assert mod.cfg["num_latents"] == 100 and low.num_latents == 100
    kinds = [o.kind for o in low.ops]
    assert kinds == [1, 1, 1, 1, 4, 4, 4, 2, 2, 2, 2]
    assert low.zpre_slot == low.z_slot and not low.has_made
    names = [o.name for o in low.ops]
    assert names == ["enc_conv1", "enc_conv2", "enc_conv3", "enc_conv4", "enc_fc1", "enc_mu", "l_dec_fc2",
                     "dec_conv1", "dec_conv2", "dec_conv3", "dec_out"]
    fc1, fc2 = low.ops[4], low.ops[6]
    assert fc1.flat == (1024, 4, 4) and fc1.act == lowering.ACTS["elu"] and fc1.bn_name == "bnorm_enc_fc1"
    assert fc2.unflat == (1024, 4, 4) and fc2.act == lowering.ACTS["relu"] and fc2.has_bias == 0
    assert low.ops[0].has_bias == 1 and low.ops[1].has_bias == 0   # batch_norm strips the bias (App. B.3)
    assert low.ops[-1].act == lowering.ACTS["tanh"] and low.ops[-1].bn_name is None
    assert low.slots[low.out_slot] == (64, 64, 3)
    # parameter inventory == the oracle's inference parameter set minus the unused log-sigma head
    want = set(O.param_shapes("IAN_simple")) - {"enc_logsigma.W"} - {"ls_bnorm." + s for s in ("beta", "gamma", "mean", "inv_std")}
    assert {p.name for p in low.params} == want
    shapes = O.param_shapes("IAN_simple")
    assert all(tuple(shapes[p.name]) == p.shape for p in low.params)


def test_ian_simple_dnn_false_branch_is_the_same_graph():
    """TransposedConv2DLayer(crop=1)+SliceLayer (IAN_simple.py:182-223) lowers to the same ops."""
    _, a = lowered(os.path.join(CFG, "IAN_simple.py"), dnn=True)
    _, b = lowered(os.path.join(CFG, "IAN_simple.py"), dnn=False)
    assert op_signature(a) == op_signature(b)


def test_ian_lowering():
    mod, low = lowered(os.path.join(CFG, "IAN.py"))
    assert low.has_made and low.zpre_slot != low.z_slot
    by_name = {o.name: o for o in low.ops}
    assert by_name["l_IAF"].kind == lowering.OP_MADE_IAF and by_name["l_IAF"].segment == lowering.SEG_IAF
    # MDBLOCK (layers.py:411-416): affine(bnorm0) -> mdc(bnorm1) -> mdc + residual(bnorm2)
    blk = [o for o in low.ops if o.name.startswith("dec_conv2a")]
    assert [o.kind for o in blk] == [lowering.OP_AFFINE, lowering.OP_MDC3, lowering.OP_MDC3]
    assert blk[0].bn_name == "dec_conv2abnorm0" and blk[0].act == lowering.ACTS["lrelu"]
    assert blk[1].bn_name == "dec_conv2abnorm1" and blk[1].src == blk[0].dst and blk[1].scales == [0, 2]
    assert blk[2].bn_name == "dec_conv2abnorm2" and blk[2].src2 == by_name["dec_conv1"].dst
    # the deconv under an MDBLOCK loses its bias to batch_norm() (App. B.3)
    assert by_name["dec_conv1"].has_bias == 0 and by_name["dec_conv1"].act == 0
    assert by_name["l_dec_fc2"].has_bias == 1 and by_name["l_dec_fc2"].unflat == (512, 4, 4)
    # RGB-Beta head (IAN.py:183-207)
    assert by_name["G_b"].src == by_name["R"].dst and by_name["G_b"].src2 == by_name["G_a"].dst
    assert by_name["B_b"].cin == 4 and by_name["B_b"].act == lowering.ACTS["sigmoid"]
    assert low.ops[-1].kind == lowering.OP_BETA
    assert {p.name for p in low.params} | {"enc_logsigma.W"} | {"ls_bnorm." + s for s in ("beta", "gamma", "mean", "inv_std")} \
        == set(O.param_shapes("IAN"))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("name,dnn", [("IAN_simple.py", True), ("IAN_simple.py", False), ("IAN.py", True)])
def test_reference_configs_load_unchanged(name, dnn):
    """The reference's own config files execute under the recorder and give the graph of our configs."""
    _, ours = lowered(os.path.join(CFG, name), dnn=dnn)
    mod, theirs = lowered(os.path.join(REF, name), dnn=dnn)
    assert op_signature(ours) == op_signature(theirs)
    assert [(p.name, p.shape) for p in ours.params] == [(p.name, p.shape) for p in theirs.params]
    assert ours.slots == theirs.slots
    ours_mod = cl.load_config(os.path.join(CFG, name))
    assert ours_mod.cfg == mod.cfg


def test_stub_environment_is_removed_afterwards():
    import sys
    cl.load_config(os.path.join(CFG, "IAN_simple.py"))
    assert "lasagne" not in sys.modules and "theano" not in sys.modules and "layers" not in sys.modules


def test_unsupported_graph_fails_loudly():
    mod = cl.load_config(os.path.join(CFG, "IAN_simple.py"))
    with cl.stub_environment():
        model = mod.get_model(dnn=True)
        import lasagne
        model["l_out"] = lasagne.layers.Conv2DLayer(model["l_out"], 3, 3, pad=1, name="extra")
    with pytest.raises(lowering.LoweringError):
        lowering.lower_model(model)


# ---- MADE masks in the product path ------------------------------------------------------------------------
def test_product_made_masks_bit_exact():
    got = made.masks_once(100)
    want = O.made_masks()
    for a, b in zip(got, want):
        assert a.dtype == np.float32 and np.array_equal(a, b)
    assert [int(m.sum()) for m in got] == [100, 9900, 4950]
    assert made.shuffled_ordering(100)[:6].tolist() == [52, 79, 87, 45, 24, 71]


# ---- checkpoint format (GANcheckpoints.py:11-57) ---------------------------------------------------------
def test_checkpoint_roundtrip_and_mismatch_policy(tmp_path):
    _, low = lowered(os.path.join(CFG, "IAN_simple.py"))
    specs = [p for p in low.params if p.name in ("enc_conv1.W", "enc_conv1.b", "bnorm2.mean")]
    rs = np.random.RandomState(0)
    arrays = {p.name: rs.randn(*p.shape).astype(np.float32) for p in specs}
    arrays["bnorm2.mean"] = np.zeros((7,), np.float32)  # wrong shape -> skipped with a warning
    f = tmp_path / "w.npz"
    checkpoints.save_weights(f, arrays, metadata={"epoch": 3, "learning_rate": 2e-4})
    checkpoints.save_weights(f, arrays, metadata={"epoch": 4, "learning_rate": 1e-4})  # overwrite via tmp + rename
    with pytest.warns(UserWarning):
        got, meta = checkpoints.load_weights(f, specs)
    assert meta == {"epoch": 4, "learning_rate": 1e-4}
    assert set(got) == {"enc_conv1.W", "enc_conv1.b"}
    assert np.array_equal(got["enc_conv1.W"], arrays["enc_conv1.W"])


def test_metadata_is_written_the_way_the_reference_reads_it(tmp_path):
    """GANcheckpoints.py:54 does ``pickle.loads(str(param_dict['metadata']))`` under Python 2: the entry must be a 0-d
    '|S' array holding a protocol-0 (ASCII) pickle of plain Python values -- no numpy._core references, which
    Python-2 numpy could not import."""
    import pickle
    import pickletools
    f = tmp_path / "w.npz"
    checkpoints.save_weights(f, {"enc_conv1.b": np.zeros(128, np.float32)},
                             metadata={"epoch": 3, "itr": 1200, "ts": 1474000000.25, "learning_rate": np.float32(2e-4), "note": "x"})
    with np.load(str(f), allow_pickle=False) as a:       # loads without pickle support: no object arrays inside
        m = a["metadata"]
        assert m.shape == () and m.dtype.kind == "S"
        raw = m.item()
    assert raw.startswith(b"(dp0\nS'epoch'\np1\nI3\ns") and raw.endswith(b"s.")   # cPickle protocol-0 text
    assert b"numpy" not in raw and max(raw) < 128
    ops = [op.name for op, _, _ in pickletools.genops(raw)]
    assert "GLOBAL" not in ops and "REDUCE" not in ops
    meta = pickle.loads(raw)                             # the reference's decode, in today's Python
    assert meta == {"epoch": 3, "itr": 1200, "ts": 1474000000.25, "learning_rate": float(np.float32(2e-4)), "note": "x"}
    assert type(meta["learning_rate"]) is float
    with pytest.raises(TypeError):
        checkpoints.save_weights(f, {}, metadata={"bad": np.zeros(3)})


def test_python2_style_checkpoint_fixture_loads():
    """tests/golden/py2_style_checkpoint.npz (made by make_py2_checkpoint.py) is laid out as GANcheckpoints.save_weights
    writes under Python 2: '|S' metadata = cPickle protocol-0 text incl. the numpy.core.multiarray.scalar reduce of
    np.float32(learning_rate) (train_IAN.py:571).  Loaded with allow_pickle=False + the restricted unpickler."""
    _, low = lowered(os.path.join(CFG, "IAN_simple.py"))
    fx = os.path.join(ROOT, "tests", "golden", "py2_style_checkpoint.npz")
    specs = lowering.all_param_specs(cl.build_model(cl.load_config(os.path.join(CFG, "IAN_simple.py"))))
    with pytest.warns(UserWarning, match="shape mismatch:enc_conv1.W"):
        got, meta = checkpoints.load_weights(fx, specs, extra_prefixes=("discrimi.", "minibatch_discrim."))
    assert meta["epoch"] == 3 and meta["itr"] == 1200 and meta["ts"] == 1474000000.25
    assert isinstance(meta["learning_rate"], np.float32) and meta["learning_rate"] == np.float32(2e-4)
    assert {"mu_bnorm.gamma", "mu_bnorm.mean", "mu_bnorm.inv_std", "enc_conv1.b", "dec_out.W", "discrimi.W"} <= set(got)
    assert "enc_conv1.W" not in got and got["dec_out.W"].shape == (128, 3, 5, 5) and got["dec_out.W"].dtype == np.float32
    rs = np.random.RandomState(7)
    assert np.array_equal(got["mu_bnorm.gamma"], rs.uniform(0.5, 1.5, 100).astype(np.float32))


def test_metadata_unpickler_refuses_code():
    import pickle
    evil = b"cos\nsystem\n(S'echo pwned'\ntR."
    with pytest.raises(pickle.UnpicklingError):
        checkpoints.loads_metadata(evil)
    with pytest.raises(pickle.UnpicklingError):
        checkpoints.loads_metadata(np.array(evil))


def test_round1_style_metadata_still_loads(tmp_path):
    """Archives written by round 1 of this repo hold the metadata as a uint8 vector containing a Python-3 protocol-2
    pickle with an np.float32 learning rate (a `_codecs.encode` global): --resume must keep working on them."""
    import pickle
    meta = {"learning_rate": np.float32(1e-3), "epoch": 7, "itr": 123}
    fname = str(tmp_path / "r1.npz")
    np.savez_compressed(fname, **{"enc_conv1.b": np.zeros(128, np.float32),
                                  "metadata": np.frombuffer(pickle.dumps(meta, protocol=2), np.uint8)})

    class Spec:
        name, shape = "enc_conv1.b", (128,)
    found, got = checkpoints.load_weights(fname, [Spec])
    assert got["epoch"] == 7 and got["itr"] == 123 and abs(float(got["learning_rate"]) - 1e-3) < 1e-9
    assert found["enc_conv1.b"].shape == (128,)


# ---- C ABI ---------------------------------------------------------------------------------------------------
def test_library_builds_loads_and_exports_header_symbols():
    lib = L.load_library()
    header = open(os.path.join(ROOT, "include", "ian.h")).read()
    declared = set(re.findall(r"\b(ian_[a-z_0-9]+)\s*\(", header))
    assert declared == set(L.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert b"gfx950" in lib.ian_version()


def test_struct_layout_matches_header():
    # ian_op_desc: 18 int32 + n_scales + 4 scales = 23 int32 (92 B) padded to 96, then two pointers
    assert ctypes.sizeof(L.OpDesc) == 96 + 16
    assert L.OpDesc.name.offset == 96
    assert ctypes.sizeof(L.SlotDesc) == 12


def test_headers_are_plain_c_and_ctypes_mirrors_their_structs(tmp_path):
    """include/ian.h and include/ian_train.h compile as C (gcc -std=c99 -pedantic: what a cgo / ctypes / JNI binding sees), and
    every ctypes.Structure in lib.py has the size and the field offsets the C compiler gives the header's struct."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    from neural_photo_editor_amd import trainer as T
    mirrors = {"ian_op_desc": L.OpDesc, "ian_slot_desc": L.SlotDesc, "ian_model_desc": L.ModelDesc, "ian_photo_args": L.PhotoArgs,
               "ian_train_config": T.TrainConfig, "ian_comm_ops": T.CommOps}     # the training entry's config and collective table
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "ian.h"', '#include "ian_train.h"', 'int main(void) {']
    for cname, cls in mirrors.items():
        lines.append('  printf("%s size %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['  return 0;', '}']
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    seen = 0
    for line in filter(None, out):
        cname, field, val = line.split()
        cls = mirrors[cname]
        if field == "size":
            assert ctypes.sizeof(cls) == int(val), (cname, ctypes.sizeof(cls), val)
        else:
            assert getattr(cls, field).offset == int(val), (cname, field)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in mirrors.values())


def test_model_creation_without_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from neural_photo_editor_amd import IAN
    with pytest.raises(L.IanError, match="no HIP device"):
        IAN(os.path.join(CFG, "IAN_simple.py"), True, params=O.make_params("IAN_simple", 1))


def test_missing_parameter_is_an_error_in_the_c_layer():
    _, low = lowered(os.path.join(CFG, "IAN_simple.py"))
    h = L.Handle(low)
    h.load_param("enc_conv1.W", np.zeros((128, 3, 5, 5), np.float32))
    with pytest.raises(L.IanError):
        h.finalize()   # either "no HIP device" (CPU box) or "missing parameter" (GPU box): never a silent success
    h.close()


def test_collective_table_callbacks_report_errors_as_return_codes():
    """trainer.build_ops (the ian_comm_ops filler): Python exceptions never cross the C boundary -- the C side sees a non-zero
    return code (and fails the step with -30), the exception is kept for the error message; arguments arrive as plain ints."""
    from neural_photo_editor_amd import trainer as T
    seen, errors = [], []

    def allreduce(buf, count, stream):
        seen.append(("ar", buf, count, stream))

    def wait_all(stream):
        raise RuntimeError("collective backend went away")

    def allgather(src, dst, count, stream):
        seen.append(("ag", src, dst, count, stream))

    ops = T.build_ops(4, 2, allreduce, wait_all, allgather, errors)
    assert (ops.world, ops.rank) == (4, 2)
    assert ops.allreduce_sum(None, 0x1000, 77, 0x2000) == 0 and seen[-1] == ("ar", 0x1000, 77, 0x2000)
    assert ops.allgather(None, 0x10, 0x20, 5, None) == 0 and seen[-1] == ("ag", 0x10, 0x20, 5, None)
    assert ops.wait_all(None, 0) == 1 and len(errors) == 1 and "went away" in str(errors[0])


def test_training_abi_exports_every_header_symbol():
    """include/ian_train.h: every declared entry point resolves in libian.so and gets its prototype from the header."""
    lib = L.load_train_library()
    names = L.train_exports()
    assert len(names) >= 39 and "ian_layer_backward_weight" in names and "ian_k_adam" in names
    for n in ("ian_train_step", "ian_trainer_set_comm", "ian_trainer_forward", "ian_trainer_backward", "ian_trainer_finish_allreduce",
              "ian_trainer_buffer", "ian_k_axpy_f64"):
        assert n in names, n
    for n in names:
        assert getattr(lib, n).argtypes is not None


def test_layer_creation_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from neural_photo_editor_amd.trainer import Layer, IanTrainError
    with pytest.raises(IanTrainError, match="no HIP device"):
        Layer(L.load_train_library(), 4, 128, 64)


def test_host_class_has_the_reference_surface():
    """Drop-in boundary: every public method of the reference's plat-style class (API.py:11-110) exists on
    neural_photo_editor_amd.IAN with the same positional argument names, and the constructor takes
    (config_path, dnn).  Reads the reference source with ast (Python 2 syntax is not executed)."""
    import inspect
    ref = os.path.join(REF, "API.py")
    if not os.path.exists(ref):
        pytest.skip("reference checkout not present")
    import ast
    src = open(ref).read().replace("print ", "pass #")     # py2 print statements
    tree = ast.parse(src)
    cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "IAN"][0]
    ref_methods = {f.name: [a.arg for a in f.args.args] for f in cls.body if isinstance(f, ast.FunctionDef)}
    assert set(ref_methods) == {"__init__", "imgrad", "imgradRGB", "encode_images", "get_zdim", "sample_at"}
    from neural_photo_editor_amd import IAN
    for name, args in ref_methods.items():
        ours = list(inspect.signature(getattr(IAN, name)).parameters)
        assert ours[:len(args)] == args, (name, ours, args)
    # what NPE.py calls (NPE.py:18,110,205,218,257,261,296,311,323,333,335)
    npe = open(os.path.join(REF, "NPE.py")).read()
    used = set(re.findall(r"model\.(\w+)\(", npe))
    assert used <= set(ref_methods) and "imgradRGB" in used and "sample_at" in used


def test_bench_quotes_pmc_traffic_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    """bench.py's roofline.traffic comes from a committed rocprofv3 summary (PMC counters cannot be read in-process): it may be
    quoted only when the summary's csrc digest is the digest of the sources the running library was built from; otherwise the
    field is null and the reason names both digests."""
    import json
    import bench
    from neural_photo_editor_amd import build
    for arch, B, tag in (("IAN_simple", 64, "ian_simple_b64"), ("IAN", 256, "ian_b256")):
        v, src = bench.pmc_traffic(arch, B)           # the newest committed summary: quoted if it is this build's, else labelled
        newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_%s.json" % tag))[-1]
        fresh = json.load(open(os.path.join(ROOT, "profiles", newest))).get("csrc_digest") in (build._digest("inference"), build._digest())
        assert newest in src and ((v and v > 1e6) if fresh else (v is None and "stale" in src)), (v, src, fresh)
    assert bench.pmc_traffic("IAN_simple", 32) == (None, None)            # no committed profile for that workload
    (tmp_path / "profiles").mkdir()
    stale = {"tapgemm_traffic_bytes_per_launch": 1.0e8, "csrc_digest": "0" * 64}
    (tmp_path / "profiles" / "r09_ian_simple_b64.json").write_text(json.dumps(stale))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    v, why = bench.pmc_traffic("IAN_simple", 64)
    assert v is None and "stale" in why and build._digest("inference")[:12] in why
    stale["csrc_digest"] = build._digest("inference")     # training-only sources (kernels_train.hip, ian_trainer.cpp ...) do not enter
    (tmp_path / "profiles" / "r09_ian_simple_b64.json").write_text(json.dumps(stale))
    assert bench.pmc_traffic("IAN_simple", 64) == (1.0e8, os.path.join("profiles", "r09_ian_simple_b64.json"))
