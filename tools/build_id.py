"""Identity of the kernel sources a measurement belongs to: sha256 over csrc/*.hip, csrc/*.h, csrc/Makefile and include/ttk.h (sorted by
name).  It is computed from file CONTENTS, so it is the same in the build container and on a GPU box (whose snapshot has no .git).
tools/pmc_summary.py stamps it into its JSON; bench.py attaches counter traffic only from a summary whose stamp equals the running tree's.
  python tools/build_id.py            -> prints the hash
  python tools/build_id.py --stamp-head <summary.json>   (build container: adds the git HEAD the summary was committed beside)"""
import glob, hashlib, json, os, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256() -> str:
    h = hashlib.sha256()
    c = os.path.join(REPO, "neuralnet-tracker-traincode_amd", "csrc")
    files = sorted(glob.glob(os.path.join(c, "*.hip")) + glob.glob(os.path.join(c, "*.h")) + [os.path.join(c, "Makefile"), os.path.join(REPO, "include", "ttk.h")])
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--stamp-head":
        d = json.load(open(sys.argv[2]))
        d["git_head"] = subprocess.check_output(["git", "-C", REPO, "rev-parse", "HEAD"], text=True).strip()
        d["git_head_note"] = "HEAD of the build container when the summary was copied into profiles/ (the commit that holds it is its child)"
        json.dump(d, open(sys.argv[2], "w"), indent=1)
        print(d["git_head"], d.get("csrc_sha256"))
    else:
        print(csrc_sha256())
