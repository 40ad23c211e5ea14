"""Which GPU is this?  The pool's containers all share one hostname, so evidence files name the box by the GPU's
`unique_id` from the KFD topology (/sys/class/kfd/kfd/topology/nodes/*/properties) -- readable without touching the HIP
runtime.  Used by bench.py (roofline.traffic_source) and tools/pmc_traffic.py."""
import os

_ROOT = "/sys/class/kfd/kfd/topology/nodes"


def gpu_unique_ids():
    """unique_id of every KFD node that has SIMDs, in node order (= HIP device order when no visibility mask reorders them)."""
    out = []
    try:
        nodes = sorted(os.listdir(_ROOT), key=lambda s: int(s) if s.isdigit() else 1 << 30)
    except OSError:
        return out
    for node in nodes:
        try:
            props = dict(line.split() for line in open(os.path.join(_ROOT, node, "properties")) if len(line.split()) == 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            out.append(props.get("unique_id", "unknown"))
    return out


def gpu_unique_id(index=0):
    ids = gpu_unique_ids()
    return ids[index] if 0 <= index < len(ids) else "unknown"
