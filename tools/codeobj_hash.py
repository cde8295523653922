#!/usr/bin/env python3
"""SHA-256 of the DISASSEMBLY (instruction text without addresses / encodings) of every gfx950 code object inside a librmd_hip.so: two builds
whose device code is the same print the same hashes whatever happened to comments, line breaks or file boundaries of the sources.
usage: python tools/codeobj_hash.py [library=rpg_open_remode_amd/librmd_hip.so]"""
import hashlib, os, re, shutil, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rpg_open_remode_amd", "librmd_hip.so")
d = tempfile.mkdtemp(prefix="rmd_co_")
try:
    shutil.copy(lib, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(d, f)], capture_output=True, text=True, check=True).stdout
        lines = []
        for line in dis.splitlines():
            line = re.sub(r"//.*$", "", line)                      # the trailing address comment
            line = re.sub(r"^\s*[0-9a-f]+:\s*", "", line)           # a leading address
            line = re.sub(r"^[0-9a-f]{8,} <(.*)>:$", r"<\1>:", line)  # symbol headers without their address
            if line.strip() and "file format" not in line:  # (the header names the temporary file)
                lines.append(line.strip())
        kernels = sorted(set(re.findall(r"<(_Z[^>]+)>:", "\n".join(lines))))
        print(f.split("hipv4-")[-1][-60:], hashlib.sha256("\n".join(lines).encode()).hexdigest()[:16], f"{len(lines)} lines, {len(kernels)} symbols")
finally:
    shutil.rmtree(d, ignore_errors=True)
