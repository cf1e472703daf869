"""Who owns the descriptor handed to hipImportExternalMemory? Exports a device allocation as a dma-buf, imports a dup of the fd through
vt_import_dmabuf and lists this process's open descriptors before / after the import and after the release (run on the GPU box)."""
import os
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import torch
import gstreamer_vit_tracker_amd as vt


def fds():
    out = {}
    for n in os.listdir("/proc/self/fd"):
        try:
            out[int(n)] = os.readlink(f"/proc/self/fd/{n}")
        except OSError:
            pass
    return out


torch.cuda.empty_cache()
buf = torch.zeros(32 << 20, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
fd = vt.export_dmabuf(buf.data_ptr(), 32 << 20)
a = fds()
print("exported fd", fd, "->", a.get(fd))
m = vt.DmaBuf(fd, 32 << 20)
b = fds()
print("after import : new fds", {k: v for k, v in b.items() if k not in a}, " gone", [k for k in a if k not in b])
m.close()
c = fds()
print("after release: new vs before import", {k: v for k, v in c.items() if k not in a}, " gone vs after import", [k for k in b if k not in c])
assert set(c) == set(a), "vt_release_dmabuf left a descriptor behind"
os.close(fd)
print("done")
