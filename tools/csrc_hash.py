"""sha256 over the kernels' sources (hugectr_amd/csrc/*.hip, *.h, *.cpp): counter files under
profiles/ are stamped with it, and bench.py quotes a counter figure only while the hash still
matches the sources it runs (a box without .git can check this, a commit id cannot be)"""
import glob
import hashlib
import os


def csrc_hash(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "hugectr_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) +
                    glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_hash())
