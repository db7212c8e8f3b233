"""Dev tool: sha256 over the grid search kernel's sources (icp_grid.hip + icp_grid_device.h) -- the tag the PMC collection
scripts embed in profiles/pmc_traffic.json / pmc_issue.json and bench.py compares with the tree it runs from (`pmc_stale`)."""
import hashlib, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def search_kernel_source_sha256() -> str:
    h = hashlib.sha256()
    for f in ("icp_grid.hip", "icp_grid_device.h"):
        with open(os.path.join(ROOT, "icpslam_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


if __name__ == "__main__":
    print(search_kernel_source_sha256())
