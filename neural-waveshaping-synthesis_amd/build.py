"""Build libnws_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain C-ABI.
Also libnws_probe.so: the MI355X co-execution hazard probe (csrc/coexec_probe.hip) - tools and tests only, kept OUT of the
product library because its kernels contain, on purpose, the instruction form the build guard refuses everywhere else.

    python neural-waveshaping-synthesis_amd/build.py [--force]
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libnws_hip.so")
OPS_LIB = os.path.join(HERE, "libnws_torch_ops.so")     # torch.ops.newt_hip.* over the C-ABI (csrc/torch_ops.cpp)
PROBE_LIB = os.path.join(HERE, "libnws_probe.so")       # include/nws_probe.h: hazard probe, tools / tests only
SOURCES = ["exciter_newt.hip", "control_gru.hip", "frame_mlps.hip", "fir_noise.hip", "reverb_fft.hip", "forward.hip",
           "loudness.hip", "stages.hip", "generic.hip", "stream.hip", "queue_probe.hip", "exchange.hip"]
PROBE_SOURCES = ["coexec_probe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=default",
         "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-Rpass-analysis=kernel-resource-usage"]
# -fno-slp-vectorize: the SLP vectoriser turns scalar fp32 code into packed instructions with operand swizzles of its own
# choosing, among them the form that misbehaves beside another kernel's f16 MFMAs (see check_packed_swizzles below);
# packed arithmetic is written explicitly (f32x2) where it pays.  Two files keep the vectoriser (the guard checks every
# file anyway) because an MFMA kernel in each sits at its register budget and spills without it: fir_noise.hip (only
# receives the harmless low-broadcast form) and control_gru.hip (batched kernel; the per-utterance kernel's horizontal
# sums are scalar by hand).
KEEP_SLP = ("fir_noise.hip", "control_gru.hip")
EXTRA_FLAGS = {src: ["-fno-slp-vectorize"] for src in SOURCES + PROBE_SOURCES if src not in KEEP_SLP}
# experiments: extra hipcc flags for every file (e.g. NWS_EXTRA_HIPCC_FLAGS="-DNWS_EXCITER_PRIO_MIX=1"); part of the build stamp
FLAGS += [f for f in os.environ.get("NWS_EXTRA_HIPCC_FLAGS", "").split() if f]
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
# kernels that contain the hazardous form on purpose (the probe that demonstrates it): allowed in PROBE_SOURCES only - an object
# of the product library is checked with an empty allow-list
SWIZZLE_ALLOW = ("pk_probe_kernel", "pk_probe2_kernel", "pk_probe_mixed_kernel")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp():
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(EXTRA_FLAGS.items()))).encode())
    for f in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and f != "torch_ops.cpp":
            h.update(f.encode())
            h.update(open(p, "rb").read())
    for hdr in ("nws_hip.h", "nws_hip_debug.h", "nws_probe.h"):
        h.update(open(os.path.join(HERE, "..", "include", hdr), "rb").read())
    return h.hexdigest()


def device_disassembly(obj):
    """gfx950 ISA of the device code embedded in a host object built by hipcc."""
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        r = subprocess.run([os.path.join(LLVM_BIN, "llvm-objcopy"), "--dump-section", f".hip_fatbin={fat}", obj,
                            os.path.join(td, "host.o")], capture_output=True, text=True)
        if r.returncode != 0:
            if "not found" in r.stderr:      # a host-only translation unit (no kernels)
                return ""
            raise RuntimeError(f"llvm-objcopy failed on {obj}: {r.stderr}")
        subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", co], check=True, capture_output=True,
                              text=True).stdout


def check_packed_swizzles(obj, allow=None):
    """MI355X co-execution hazard guard (LABBOOK.md '5.2', csrc/coexec_probe.hip).

    v_pk_{add,mul,fma}_f32 with op_sel[1] = 1 (low lane <- high half of src1) returns wrong values while another kernel
    runs K=16/32 f16 MFMAs on the same CU.  The check is wider than what was seen to fail: ANY vector instruction with
    op_sel[1] = 1 is refused (packed fp16, v_fma_mix* and scalar-register operands probed clean; no product kernel needs
    those forms either, so the guard stays simple).
    `allow`: kernel-name substrings that may contain the form (default SWIZZLE_ALLOW; the product objects pass ()).
    Returns [(kernel, instruction), ...] for every other occurrence."""
    import re
    allow = SWIZZLE_ALLOW if allow is None else allow
    found, kernel = [], None
    for line in device_disassembly(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        m = re.search(r"\b(v_\w+)\b.*?op_sel:\[\d,(\d)", line)
        if m and m.group(2) == "1" and not any(a in (kernel or "") for a in allow):
            found.append((kernel, " ".join(line.split("//")[0].split())))
    return found


def _vgprs(tok):
    """'v[4:7]' -> {4..7}, 'v9' -> {9}; anything else (sgprs, constants, agprs) -> empty"""
    import re
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check_valu_mfma_hazard(obj, wait_states=2):
    """gfx940-class rule: a VGPR written by a (non-matrix) VALU instruction must not be read by a v_mfma within the next
    `wait_states` wait states.  hipcc places those itself for the instructions it schedules, but it does not see through INLINE
    ASM: round 4 found a layer's first MFMA reading a stale B operand that an inline-asm v_fma_mixhi_f16 had written two
    instructions earlier (csrc/frame_mlps.hip, wr_split2).  Every v_mfma in the device code is checked against the
    instructions in front of it (s_nop N counts N + 1 wait states; branch targets end the look-back).
    Returns [(kernel, producer, mfma), ...]."""
    import re
    found, kernel, window = [], None, []      # window: [(wait states this instruction accounts for, dest vgprs, text)]
    for line in device_disassembly(obj).splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel, window = m.group(1), []
            continue
        parts = line.split("//")[0].split("\t")
        text = " ".join(" ".join(parts[1:]).split()) if len(parts) > 1 else ""
        if not text:
            continue
        op = text.split()[0]
        ops = [t.strip() for t in text[len(op):].split(",")]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            srcs = set().union(*[_vgprs(t.split()[0]) for t in ops[1:4] if t])
            left = wait_states
            for ws, dst, ptext in reversed(window):
                if left <= 0:
                    break
                if dst & srcs:
                    found.append((kernel, ptext, text))
                left -= ws
        if op == "s_nop":
            window.append((int(ops[0], 0) + 1, set(), text))
        elif op.startswith("s_cbranch") or op in ("s_branch", "s_barrier", "s_endpgm", "s_setpc_b64"):
            window = []
        elif op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_smfmac") and not op.startswith("v_cmp"):
            window.append((1, _vgprs(ops[0].split()[0]) if ops and ops[0] else set(), text))
        else:
            window.append((1, set(), text))
        window = window[-8:]
    return found


# Register budgets that decide how many workgroups a CU holds (round 4: the oscillator kernel at 93 VGPRs ran TWO 8-wave
# workgroups per CU although the resource remark said "5 waves per SIMD" - a workgroup brings 2 waves per SIMD, so the steps are
# 4 / 6 / 8 waves = 128 / 80 / 64 registers; the third workgroup was worth 3-6 %).  (substring of the mangled kernel name,
# largest VGPR + AGPR count, what it buys); a kernel that outgrows its line fails the build instead of silently losing occupancy.
REGISTER_BUDGETS = (
    ("exciter_newt_kernelILi4ELi0ELi2ELi34E", 80, "default oscillator kernel: three 8-wave workgroups per CU"),
    ("exciter_newt_kernelILi4ELi0ELi2ELi38E", 80, "one-term variant: three 8-wave workgroups per CU"),
    ("exciter_newt_kernelILi4ELi0ELi2ELi42E", 80, "hybrid variant: three 8-wave workgroups per CU"),
    ("exciter_newt_kernelILi4ELi0ELi2ELi58E", 80, "hybrid-W variant: three 8-wave workgroups per CU"),
    ("exciter_newt_kernelILi5ELi0ELi1ELi0E", 128, "exact-shaper bank kernel: four 4-wave workgroups per CU (38.8 KB of LDS each)"),
    ("exciter_newt_kernelILi6ELi0ELi1ELi0E", 128, "exact-shaper bank kernel (no v_fract): four 4-wave workgroups per CU"),
    ("g_exciter_newt_mfma_kernel", 128, "runtime-size oscillator kernel: four 4-wave workgroups per CU"),
)


def kernel_registers(remarks):
    """{mangled kernel name: VGPRs + AGPRs} from hipcc's -Rpass-analysis=kernel-resource-usage output"""
    regs, name = {}, None
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[-Rpass")[0].strip()
            regs[name] = 0
        elif name is not None and (" VGPRs:" in line or " AGPRs:" in line) and "Spill" not in line:
            regs[name] += int(line.split("GPRs:")[1].split()[0])
    return regs


def check_register_budgets(remarks, budgets=None):
    """[(kernel, registers, budget, reason), ...] for every kernel of `remarks` that exceeds its line of REGISTER_BUDGETS"""
    over = []
    for kernel, n in kernel_registers(remarks).items():
        for key, budget, why in (REGISTER_BUDGETS if budgets is None else budgets):
            if key in kernel and n > budget:
                over.append((kernel, n, budget, why))
    return over


def build_torch_ops(force=False, verbose=True):
    """torch.ops.newt_hip.*: one host-only translation unit (no kernels) compiled with g++ against the torch headers and linked
    to libnws_hip.so next to it.  Needs an importable torch; returns None (with a note) when there is none - the ctypes
    binding of the C-ABI keeps working without it."""
    try:
        import torch
        from torch.utils import cpp_extension as ce
    except Exception as e:   # pragma: no cover
        print(f"torch not importable ({e}): skipping libnws_torch_ops.so", file=sys.stderr)
        return None
    src = os.path.join(CSRC, "torch_ops.cpp")
    h = hashlib.sha256(open(src, "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "nws_hip.h"), "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "nws_hip_debug.h"), "rb").read())
    h.update(torch.__version__.encode())
    stamp, stamp_file = h.hexdigest(), OPS_LIB + ".stamp"
    if not force and os.path.exists(OPS_LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return OPS_LIB
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-DUSE_ROCM",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wall", "-Wno-unused-function",
           *[f"-I{p}" for p in ce.include_paths()], "-I/opt/rocm/include", src, "-o", OPS_LIB,
           f"-L{HERE}", "-lnws_hip", f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
           "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"g++ failed on torch_ops.cpp:\n{r.stderr[-6000:]}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print(f"built {OPS_LIB}")
    return OPS_LIB


def build(force=False, verbose=True):
    lib = build_hip(force, verbose)
    build_torch_ops(force, verbose)
    return lib


def build_hip(force=False, verbose=True):
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if (not force and os.path.exists(LIB) and os.path.exists(PROBE_LIB) and os.path.exists(stamp_file)
            and open(stamp_file).read() == stamp):
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        # No kernel may use scratch (spills or a device-call stack frame): every occurrence so far was a register-allocation
        # accident that cost speed.  The resource remarks make it a build error.
        name, spilled, rest, skip = None, [], [], 0
        for line in r.stderr.splitlines():
            if skip and (line.lstrip().startswith("|") or line.lstrip().split(" ")[0].isdigit()):
                skip -= 1          # source line + caret that clang prints under every remark
                continue
            skip = 0
            if "remark:" in line:
                skip = 2
                if "Function Name:" in line:
                    name = line.split("Function Name:")[1].split("[-Rpass")[0].strip()
                elif "ScratchSize [bytes/lane]:" in line and int(line.split("ScratchSize [bytes/lane]:")[1].split()[0]) != 0:
                    spilled.append(name)
            else:
                rest.append(line)
        if spilled:
            # NWS_BUILD_ALLOW_SCRATCH=1: measurements of an experimental variant only (prints the list, never set by the package)
            if os.environ.get("NWS_BUILD_ALLOW_SCRATCH") == "1":
                print(f"{src}: kernels using scratch memory (allowed by NWS_BUILD_ALLOW_SCRATCH): {spilled}", file=sys.stderr)
            else:
                raise RuntimeError(f"{src}: kernels using scratch memory (spills or stack): {spilled}")
        over = check_register_budgets(r.stderr)
        if over:
            lines = "\n".join(f"  {k}: {n} registers > {b} ({why})" for k, n, b, why in over)
            raise RuntimeError(f"{src}: kernels over their register budget (REGISTER_BUDGETS: workgroups per CU):\n{lines}")
        swz = check_packed_swizzles(obj, SWIZZLE_ALLOW if src in PROBE_SOURCES else ())
        if swz:
            lines = "\n".join(f"  {k}: {i}" for k, i in swz[:12])
            raise RuntimeError(f"{src}: packed fp32 instructions with a swizzled src1 low lane (co-execution hazard, see "
                               f"check_packed_swizzles):\n{lines}")
        haz = check_valu_mfma_hazard(obj)
        if haz:
            lines = "\n".join(f"  {k}: {p}  ->  {m}" for k, p, m in haz[:12])
            raise RuntimeError(f"{src}: v_mfma reads a VGPR that a VALU instruction wrote less than 2 wait states earlier (inline asm "
                               f"in front of a matrix instruction? see check_valu_mfma_hazard):\n{lines}")
        if verbose and "\n".join(rest).strip():
            print("\n".join(rest), file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=len(SOURCES) + len(PROBE_SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES + PROBE_SOURCES))
    for lib, these in ((LIB, objs[:len(SOURCES)]), (PROBE_LIB, objs[len(SOURCES):])):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *these, "-o", lib], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
