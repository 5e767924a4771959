"""Build libnws_hip.so (gfx950) in-tree with hipcc.  No torch headers: the library is a plain C-ABI.

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
SOURCES = ["exciter_newt.hip", "control_gru.hip", "frame_mlps.hip", "fir_noise.hip", "reverb_fft.hip", "forward.hip",
           "loudness.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=default",
         "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "nws_hip.h"), "rb").read())
    return h.hexdigest()


def build(force=False, verbose=True):
    stamp_file = LIB + ".stamp"
    stamp = _stamp()
    if not force and os.path.exists(LIB) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
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
            raise RuntimeError(f"{src}: kernels using scratch memory (spills or stack): {spilled}")
        if verbose and "\n".join(rest).strip():
            print("\n".join(rest), file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp_file, "w") as f:
        f.write(stamp)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
