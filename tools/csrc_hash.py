"""SHA-256 over the kernel sources (csrc/*.hip, *.hpp, Makefile, in name order): stamps profiles/traffic_latest.json so that
bench.py can tell whether the committed PMC traffic belongs to the build it is timing."""
import hashlib, os, sys


def csrc_hash(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "fast_solver_lippmann_schwinger_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")) or f == "Makefile":
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_hash(sys.argv[1] if len(sys.argv) > 1 else None))
