"""No-GPU tier: the C-ABI library loads and exports every symbol include/turbo_metrics_hip.h (the facade a binder needs) and
include/turbo_metrics_hip_debug.h (the laboratory) declare;
host-only entry points behave; nothing here computes on a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O
from tm_pkg import tm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="turbo_metrics_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    L = tm.ffi.lib()
    facade, lab = declared_symbols(), declared_symbols("turbo_metrics_hip_debug.h")
    # the facade is what INTEGRATION.md binds: small, and free of variants / tuning / read-back hooks
    assert 30 <= len(facade) <= 43, len(facade)
    assert not [n for n in facade if "debug" in n or "variant" in n or "profiling" in n or "stage_ms" in n]
    assert not set(facade) & set(lab)
    src = open(os.path.join(ROOT, "include", "turbo_metrics_hip.h")).read()
    assert "TM_DBG_" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S) and "TM_VARIANT_" not in re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(facade + lab)
    for n in names:
        assert hasattr(L, n), f"{n} declared in the header but not exported"
        assert n in tm.ffi.SYMBOLS, f"{n} has no ctypes prototype"
    assert sorted(tm.ffi.SYMBOLS) == names


def exported(lib):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    return sorted(m.group(2) for m in re.finditer(r" ([A-Za-z]) (\S+)", out) if m.group(1) in "TDBRW" and not m.group(2).startswith(("_init", "_fini", "__bss", "_edata", "_end")))


def device_code(lib, tmp):
    """{kernel: its disassembly, addresses stripped} of the gfx950 code object inside a library"""
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin/"
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
    subprocess.check_call([llvm + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([llvm + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    txt = subprocess.run([llvm + "llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", co], capture_output=True, text=True, check=True).stdout
    kernels, name = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]* ?<(\S+)>:$", line.strip())
        if m:
            name = m.group(1); kernels[name] = []
        elif name and line.strip():
            kernels[name].append(re.sub(r"\s*//.*$", "", line.strip()))
    for k, ins in kernels.items():  # what follows the last s_endpgm is alignment padding up to the next symbol
        ends = [i for i, t in enumerate(ins) if t.startswith("s_endpgm")]
        kernels[k] = ins[:ends[-1] + 1] if ends else ins
    return kernels


def test_the_ship_library_is_the_facade_and_shares_its_kernels_with_the_laboratory_build(tmp_path):
    """`make ship` -> libturbometrics_hip.so exports exactly the entry points of include/turbo_metrics_hip.h (ship.map), no
    tm_engine_debug_*, no variants, no fault-injection or read-back hooks, and carries none of the straight-line reference kernels; `make
    lab` -> lab/libturbometrics_hip_lab.so (what the tests load) has both headers.  Every kernel the ship library carries is in the
    laboratory build too, instruction for instruction -- what the GPU tier verifies is what ships."""
    ship, lab = tm.ffi.SHIP_LIB_PATH, tm.ffi.LIB_PATH
    facade, labsyms = declared_symbols(), declared_symbols("turbo_metrics_hip_debug.h")
    assert exported(ship) == facade
    assert [n for n in exported(lab) if n.startswith("tm_")] == sorted(facade + labsyms)  # (the laboratory build has no export map: kernel stubs show too)
    mp = open(os.path.join(ROOT, "turbo-metrics_amd", "csrc", "ship.map")).read()
    assert sorted(re.findall(r"^\s+(tm_[a-z0-9_]+);", mp, flags=re.M)) == facade
    d1, d2 = tmp_path / "ship", tmp_path / "lab"
    d1.mkdir(); d2.mkdir()
    ks, kl = device_code(ship, str(d1)), device_code(lab, str(d2))
    names = lambda d: sorted(n for n in d if n.startswith("_ZN3tmk"))
    assert names(ks) and set(names(ks)) < set(names(kl))
    only_lab = sorted(set(names(kl)) - set(names(ks)))
    assert all(any(k in n for k in ("8k_ingestE", "11k_downscale", "5k_xyb", "8k_blur_vE", "13k_blur_h_jobsE")) for n in only_lab), only_lab  # the reference pipeline, nothing else
    assert len(only_lab) == 5
    for n in names(ks):
        assert ks[n] == kl[n], n
        assert len(ks[n]) > 10
    # the C client of the header links against the ship library (test_header_is_plain_c_and_host_entry_points_work) and so does the CLI
    import subprocess
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "turbo-metrics_amd", "bin", "turbo-metrics")], capture_output=True, text=True, check=True).stdout
    assert "libturbometrics_hip.so" in needed and "_lab" not in needed


def test_the_rust_binding_in_integration_md_is_the_facade_header():
    """INTEGRATION.md section 2 shows the `extern "C"` block a maintainer adds on the reference's side: exactly the entry points of
    include/turbo_metrics_hip.h -- none missing, none of the laboratory header's"""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = txt[txt.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    bound = sorted(set(re.findall(r"pub fn (tm_[a-z0-9_]+)\(", block)))
    assert bound == declared_symbols(), (sorted(set(declared_symbols()) - set(bound)), sorted(set(bound) - set(declared_symbols())))


def test_the_collective_library_exports_its_header():
    """libturbometrics_rccl.so (the ONE reduce of `turbo-metrics --ranks N`) exports exactly include/turbo_metrics_comm.h, the Rust block of
    INTEGRATION.md section 5.1 binds the same five names, and the engine library does not depend on RCCL.  Loading only: no communicator
    without a GPU."""
    import subprocess
    want = declared_symbols("turbo_metrics_comm.h")
    assert want == ["tm_comm_destroy", "tm_comm_get_unique_id", "tm_comm_init", "tm_comm_last_error", "tm_comm_reduce_sum_f64"]
    lib = os.path.join(ROOT, "turbo-metrics_amd", "libturbometrics_rccl.so")
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    assert sorted(n for n in re.findall(r" T (\w+)", out) if n.startswith("tm_")) == want
    L = C.CDLL(lib)
    L.tm_comm_last_error.restype = C.c_char_p
    assert L.tm_comm_get_unique_id(None) != 0 and b"null pointer" in L.tm_comm_last_error()
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = txt[txt.index("pub fn tm_comm_get_unique_id") - 40:]
    assert sorted(set(re.findall(r"pub fn (tm_comm_[a-z0-9_]+)\(", block[:block.index("\n}\n")]))) == want
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "turbo-metrics_amd", "libturbometrics_hip.so")], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "turbo-metrics_amd", "bin", "turbo-metrics")], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed  # the CLI loads it at run time, for --ranks only


def test_p10_packer_follows_the_headers_sentence():
    """tm_p10_pack_rows / tm_p10_row_bytes (host code of the facade) against the numpy statement of the layout in synth.py"""
    L = tm.ffi.lib()
    rng = np.random.default_rng(1)
    for n in (1, 5, 127, 128, 129, 383, 384, 385, 960, 1000, 1920, 3840):
        pl = rng.integers(0, 1024, (5, n))
        pk = tm.synth.p10_pack_plane(pl)
        assert pk.shape[1] * 4 == L.tm_p10_row_bytes(n) == 512 * -(-n // 384)
        assert np.array_equal(tm.synth.p10_unpack_plane(pk, n), pl)
        src = np.ascontiguousarray((pl | (rng.integers(0, 64, pl.shape) << 10)).astype(np.uint16))  # bits above the tenth are dropped
        dst = np.full((5, pk.shape[1] + 3), 0xFFFFFFFF, np.uint32)
        L.tm_p10_pack_rows(src.ctypes.data, src.strides[0], n, 5, dst.ctypes.data, dst.strides[0])
        assert np.array_equal(dst[:, :pk.shape[1]], pk) and (dst[:, pk.shape[1]:] == 0xFFFFFFFF).all()


def test_strerror_and_version():
    L = tm.ffi.lib()
    assert L.tm_strerror(0) == b"ok"
    assert b"todo!()" in L.tm_strerror(tm.ffi.TM_ERR_UNSUPPORTED)
    assert L.tm_version().startswith(b"turbo-metrics-hip")


def test_host_post_processing_matches_oracle():
    rng = np.random.default_rng(7)
    for w, h in [(64, 48), (1920, 1080), (3840, 2160), (67, 35)]:
        sums = rng.random(108) * np.array([w * h / 4 ** (i // 18) for i in range(108)])
        got = tm.engine.score_from_sums(sums, w, h)
        assert got == O.score_from_sums(sums, w, h)
    assert tm.engine.score_from_sums(np.zeros(108), 8, 8) == 100.0


def test_create_rejects_bad_arguments_before_touching_the_device():
    L = tm.ffi.lib()
    h = C.c_void_p()
    assert L.tm_engine_create(C.byref(h), 0, 10, 8, 1) == tm.ffi.TM_ERR_INVALID_ARG
    assert L.tm_engine_create(C.byref(h), 10, 10, 0, 1) == tm.ffi.TM_ERR_INVALID_ARG
    assert L.tm_engine_create(C.byref(h), 10, 10, 8, 0) == tm.ffi.TM_ERR_INVALID_ARG
    assert L.tm_engine_create(C.byref(h), 10, 10, 2, 1) == tm.ffi.TM_ERR_UNSUPPORTED  # SSIM needs one 11x11 window
    assert L.tm_engine_create(C.byref(h), 175, 400, 4, 1) == tm.ffi.TM_ERR_UNSUPPORTED  # MS-SSIM: five dyadic scales
    assert h.value is None


def test_product_does_not_reference_the_oracle():
    # the oracle is test infrastructure: nothing under the package may import, link or load it
    pkg = os.path.join(ROOT, "turbo-metrics_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".inc", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "libtm_oracle" not in txt and "tm_oracle.c" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_frame_selection_matches_reference_options():
    # Options semantics of compute_all (turbo-metrics/src/lib.rs:385-404) without touching a device:
    # replay the selection logic on indices.
    def select(n, every, skip, frames):
        out, decode = [], 0
        for i in range(skip, n):
            if every > 1 and decode != 0 and decode % every != 0:
                decode += 1
                continue
            if frames > 0 and decode >= frames:
                break
            decode += 1
            out.append(i)
        return out
    assert select(10, 0, 0, 0) == list(range(10))
    assert select(10, 3, 0, 0) == [0, 3, 6, 9]
    assert select(10, 0, 2, 3) == [2, 3, 4]


def test_ssim_host_functions_match_oracle():
    rng = np.random.default_rng(9)
    for w, h in [(176, 176), (1920, 1080), (200, 300)]:
        n = [(max(w >> s, 11) - 10) * (max(h >> s, 11) - 10) for s in range(5)]
        sums = rng.random((3, 5, 2)) * np.array(n)[None, :, None]
        assert tm.engine.ssim_from_sums(sums, w, h) == O.ssim_from_sums(sums, w, h)
        assert tm.engine.msssim_from_sums(sums, w, h) == O.msssim_from_sums(sums, w, h)
    g = (C.c_float * 11)()
    tm.ffi.lib().tm_ssim_window(g)
    assert np.array_equal(np.array(g, np.float32), O.ssim_window()) and abs(sum(g) - 1.0) < 1e-6


def test_header_is_plain_c_and_host_entry_points_work(tmp_path):
    import subprocess
    for flags, lib_dir, name in (([], os.path.join(ROOT, "turbo-metrics_amd"), "turbometrics_hip"),                       # the facade header against the ship library
                                 (["-DTM_ABI_CHECK_LAB"], os.path.join(ROOT, "turbo-metrics_amd", "lab"), "turbometrics_hip_lab")):  # both headers against the laboratory build
        exe = str(tmp_path / ("abi_c_check" + ("_lab" if flags else "")))
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror"] + flags + ["-o", exe, os.path.join(ROOT, "tests", "host", "abi_c_check.c"),
                               "-L" + lib_dir, "-l" + name, "-Wl,-rpath," + lib_dir])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
        assert out.stdout.startswith("abi ok: turbo-metrics-hip")
