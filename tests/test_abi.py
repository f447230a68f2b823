"""CPU checks of the drop-in boundary: the library loads, exports every symbol the header
declares, the ctypes table covers the header, and the host mirror keeps the reference's
state_dict layout.  No compute calls (no GPU here)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dpmn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dpmn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dpmn_amd import _abi
    syms = header_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(_abi.lib, s), "libdpmn_hip.so lacks %s" % s
    assert sorted(_abi.SIGNATURES) == syms, "ctypes table and include/dpmn_hip.h disagree"
    assert _abi.lib.dpmn_abi_version() >= 1


def test_cpu_tensors_are_rejected_loudly():
    from dpmn_amd import _abi, ops
    with pytest.raises(_abi.DpmnError):
        ops.linear(torch.zeros(64, 96), torch.zeros(96, 96))


def _manifest(name):
    from helpers import load_golden
    man = {}
    for row in load_golden(name)["manifest"]:
        k, shp, dt = str(row).split("|")
        man[k] = (tuple(int(s) for s in shp.split(",")) if shp else (), dt)
    return man


def _layout(sd):
    return {k: (tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()}


def test_pgrm_state_dict_layout_matches_reference_manifest():
    from dpmn_amd.model.pgrm import PGRM
    n = 6
    args = dict(patch_size=[2] * n, embed_dim=[96] * n, depths=[1] * n, num_heads=[[6]] * n, window_size=[[2, 4, 8]] * n,
                mlp_ratio=[4.] * n, drop_rate=[0.] * n, attn_drop_rate=[0.] * n, drop_path_rate=[0.] * n)
    for tag, it, mode in (("mode0_iter0", 0, False), ("mode1_iter2", 2, True)):
        man = _manifest("pgrm_" + tag)
        sd = PGRM(iter=it, mode=mode, hidden_size=3, **args).state_dict()
        assert _layout(sd) == man
        assert list(sd.keys()) == list(man.keys())


def test_pgrm_derived_buffers_match_oracle():
    from oracle import pgrm as opgrm
    from dpmn_amd.model.pgrm import _rel_index, _shift_mask
    for ws, sh in ((2, 1), (4, 2), (8, 4)):
        m = _shift_mask(16, 64, ws, sh)
        assert torch.equal(m, opgrm.shift_mask(16, 64, ws, sh))
        tbl = torch.arange((2 * ws - 1) ** 2 * 2, dtype=torch.float32).reshape(-1, 2)
        assert torch.equal(tbl[_rel_index(ws).reshape(-1)].reshape(ws * ws, ws * ws, 2).permute(2, 0, 1),
                           opgrm.relative_bias(tbl, ws))


def test_validation_subset_names_are_unique_and_avoid_the_bookkeeping_keys():
    """main.py subset_names: the per-subset tables of TextSR.train are keyed by the validation directory's leaf name
    (super_resolution.py:286 of the reference iterates a list); equal leaves or a leaf called 'epoch' / 'score' must not collapse."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("dpmn_main", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "main.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    names = m.subset_names(["/d/test/easy", "/d/test/medium/", "/d/val/easy", "/x/epoch", "/y/score"])
    assert names == ["test_easy", "medium", "val_easy", "x_epoch", "y_score"]
    assert m.subset_names(["/d/test/easy", "/d/test/medium", "/d/test/hard"]) == ["easy", "medium", "hard"]
    assert len(set(m.subset_names(["/a/easy", "/a/easy"]))) == 2


def test_library_has_no_packed_fp32_vector_ops():
    """v_pk_fma_f32 returned wrong sums while a bf16-MFMA kernel of another stream was resident on the CU; the
    library is built with the packed-fp32 target feature off (csrc/Makefile NOPK).  Guard: disassemble every gfx950 code object of the
    shipped libdpmn_hip.so -- not one v_pk_{fma,mul,add}_f32 may be left (and the scan must have seen the MFMA kernels)."""
    import re
    import subprocess
    import tempfile
    from dpmn_amd import _abi
    llvm = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")):
        pytest.skip("ROCm llvm tools not installed")
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, _abi.LIB_PATH, os.path.join(tmp, "copy.so")], check=True)
        data = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
        assert len(starts) >= 20, "one offload bundle per translation unit expected"
        packed = mfma = 0
        for i, s in enumerate(starts):
            part, co = os.path.join(tmp, "b%d.bin" % i), os.path.join(tmp, "b%d.co" % i)
            with open(part, "wb") as f:
                f.write(data[s:starts[i + 1] if i + 1 < len(starts) else len(data)])
            subprocess.run([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
            dis = subprocess.run([llvm + "/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            packed += len(re.findall(r"v_pk_(?:fma|mul|add)_f32", dis))
            mfma += len(re.findall(r"v_mfma_f32_16x16x32_bf16", dis))
    assert mfma > 1000, "the scan did not see the bf16 MFMA kernels"
    assert packed == 0, "%d packed fp32 vector instructions in libdpmn_hip.so: the NOPK flag of csrc/Makefile is gone?" % packed
