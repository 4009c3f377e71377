#!/usr/bin/env python
"""ISA lint: every s_barrier of a library must be preceded by s_waitcnt lgkmcnt(0) if an LDS store was issued since the last such wait (gfx950: no implicit wait in
front of a raw barrier). tools/lint_barriers.py LIB.so [...]; tests/test_host_logic.py runs the same scan over both in-tree libraries."""
import os, re, struct, subprocess, sys, tempfile, time
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
def code_objects(path):
    data = open(path, "rb").read()
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        base = m.start(); off = base + 32
        for _ in range(struct.unpack_from("<Q", data, base + 24)[0]):
            o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
            triple = data[off:off + tl].decode(); off += tl
            if "gfx950" in triple and sz:
                yield data[base + o:base + o + sz]
def lint(path):
    bad, kernels, barriers = [], 0, 0
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(path):
            fn = os.path.join(d, "co.elf"); open(fn, "wb").write(co)
            txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", fn], capture_output=True, text=True, check=True).stdout
            name, pending = None, None
            for line in txt.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    name, pending = m.group(1), None; kernels += 1; continue
                ins = line.strip()
                if not ins or name is None: continue
                op = ins.split()[0]
                if op.startswith("ds_write") or op.startswith("ds_store"):
                    pending = ins
                elif op == "s_waitcnt" and "lgkmcnt(0)" in ins:
                    pending = None
                elif op == "s_barrier":
                    barriers += 1
                    if pending is not None: bad.append((name, pending))
                    pending = None      # (reported once)
    return kernels, barriers, bad
for p in sys.argv[1:]:
    t = time.time(); k, b, bad = lint(p)
    print(p.split("/")[-1], "functions", k, "barriers", b, "violations", len(bad), f"{time.time()-t:.0f}s")
    seen = set()
    for n, i in bad:
        if n not in seen: seen.add(n); print("   ", n[:110], "|", i[:60])
