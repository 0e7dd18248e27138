"""The resource table the design argues from: VGPRs / AGPRs / SGPRs / LDS / scratch / occupancy of every kernel of libckks_hip.so, taken
from the compiler's kernel-resource-usage remarks of the very objects the library is linked from (__graft_entry__.build() keeps
them beside each object).  `python tools/kernel_resources.py r06` writes profiles/r06_kernel_resources.txt; no GPU needed.
tests/test_abi_cpu.py holds the hot kernels to scratch = 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    import __graft_entry__ as g
    g.build()
    rows = g.kernel_resources()
    # a kernel instantiated in several translation units is listed once per distinct figure set
    seen, uniq = set(), []
    for r in rows:
        key = (r["kernel"], r["vgprs"], r["agprs"], r["sgprs"], r["lds"], r["scratch"], r["occupancy"])
        if key not in seen:
            seen.add(key)
            uniq.append(r)
    uniq.sort(key=lambda r: (r["file"], r["kernel"]))
    out = [f"# kernel resources of libckks_hip.so (gfx950, hipcc -O3 -Rpass-analysis=kernel-resource-usage), library digest {g.library_digest()[:16]}",
           "# occupancy = waves per SIMD the register / LDS budget allows; scratch = bytes per lane of private memory (stack objects + spills)",
           f"{'file':<18}{'kernel':<58}{'VGPR':>5}{'AGPR':>5}{'SGPR':>5}{'LDS B':>7}{'scratch':>8}{'v-spill':>8}{'occ':>4}"]
    for r in uniq:
        out.append(f"{r['file']:<18}{r['kernel'][:57]:<58}{r['vgprs']:>5}{r['agprs']:>5}{r['sgprs']:>5}{r['lds']:>7}{r['scratch']:>8}{r['vgpr_spill']:>8}{r['occupancy']:>4}")
    bad = [r for r in uniq if r["scratch"]]
    out.append(f"# {len(uniq)} kernels; with scratch memory: " + (", ".join(f"{r['kernel']} ({r['scratch']} B)" for r in bad) if bad else "none"))
    path = os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.txt")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print(path, len(uniq), "kernels;", len(bad), "with scratch")


if __name__ == "__main__":
    main()
