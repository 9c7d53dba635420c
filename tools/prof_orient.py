"""One orient-mode batch on dense content (every second frame noise) for rocprofv3: rocprofv3 --kernel-trace --stats -- python3 tools/prof_orient.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualslam_amd import capi, synth
dev="cuda:0"; n=256; rows,cols=1080,1920
ctx=capi.Context(0, torch.cuda.current_stream().cuda_stream)
p=capi.default_params(rows,cols,localize=1,orient=1)
L=capi.batch_layout(p)
frames=synth.frames_torch(n,rows,cols,device=dev,noise_every=2)
o=dict(response=torch.empty((n,rows,cols),dtype=torch.float32,device=dev), nms_mask=torch.empty((n,rows,cols),dtype=torch.uint8,device=dev),
 harris_kps=torch.empty((n,p.harris_cap,3),dtype=torch.int32,device=dev), harris_counts=torch.zeros(n,dtype=torch.int32,device=dev),
 pyramid=torch.empty((n,L.pyramid_frame_bytes),dtype=torch.uint8,device=dev), extrema_bits=torch.empty((n,L.bits_frame_words),dtype=torch.int64,device=dev),
 dog_points=torch.empty((n,p.dog_cap,6),dtype=torch.int32,device=dev), dog_counts=torch.zeros(n,dtype=torch.int32,device=dev),
 oriented_points=torch.empty((n,p.oriented_cap,6),dtype=torch.int32,device=dev), oriented_counts=torch.zeros(n,dtype=torch.int32,device=dev),
 oriented_survivors=torch.zeros(n,dtype=torch.int32,device=dev))
if os.environ.get("DESCRIBE"):
    o["descriptors"]=torch.empty((n,p.oriented_cap,128),dtype=torch.float32,device=dev); o["descriptor_defined"]=torch.zeros((n,p.oriented_cap),dtype=torch.uint8,device=dev)
for i in range(3): ctx.detect_batch(p,frames,**o)
torch.cuda.synchronize()
print("survivors", int(o["oriented_survivors"].sum()), "oriented", int(o["oriented_counts"].sum()), "dog", int(o["dog_counts"].sum()))
for name in (("k_orient_survivors", "k_sift_descriptors") if os.environ.get("DESCRIBE") else ("k_orient_survivors",)):
    ctx.kernel_timing_enable(name)
    for i in range(3): ctx.detect_batch(p,frames,**o)
    torch.cuda.synchronize()
    la, ms = ctx.kernel_timing_read()
    ctx.kernel_timing_enable(None)
    print(f"{name}: {la} launches, {ms / 3:.3f} ms per step", flush=True)
