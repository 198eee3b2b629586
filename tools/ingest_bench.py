"""Times vhap_frame_ingest (gather + composite + convert of one 16 x 512 x 512 batch from a resident uint8 sequence) against its
algorithmic traffic: 4 B read + 16 B written per pixel."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from vhap_amd.ingest import FrameStore
rng = np.random.default_rng(0)
N,H,W=64,512,512
st = FrameStore(rng.integers(0,256,(N,H,W,3),dtype=np.uint8), rng.integers(0,256,(N,H,W),dtype=np.uint8), "white")
idx = torch.arange(16, device="cuda")*3
out = torch.empty(16,3,H,W,device="cuda"); a = torch.empty(16,1,H,W,device="cuda")
for _ in range(5): st.batch(idx,out=out,alpha_out=a)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): st.batch(idx,out=out,alpha_out=a)
e1.record(); torch.cuda.synchronize()
us=e0.elapsed_time(e1)*10
byt=16*H*W*(4+16)
print(f"ingest 16x512x512: {us:.1f} us/launch, {byt/us/1e3:.1f} GB/s")
