import sys, numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
from oracle import vit_ref as R
w,h=64,48
rng=np.random.default_rng(w*h); n=w*h+w*((h+1)//2)+2; buf=rng.integers(0,256,n,dtype=np.uint8)
ref,_=R.nv12_to_rgb8(buf,w,h,1); got=vt.nv12_full_to_rgb(buf,w,h)
bad=np.argwhere((ref!=got).any(axis=2))
print("n bad", len(bad), "of", w*h)
print("col%4 histogram", np.bincount(bad[:,1]%4, minlength=4))
print("chan mismatch counts", (ref!=got).reshape(-1,3).sum(axis=0))
for (r,c) in bad[:8]:
    print(r,c,"Y",buf[r*w+c],"UV",buf[w*h+(r//2)*w+(c&~1)], buf[w*h+(r//2)*w+(c&~1)+1],"ref",ref[r,c],"got",got[r,c])
