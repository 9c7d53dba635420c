"""Ad-hoc parity sweep at sizes the test-suite does not reach (4K, auto octave count, odd and portrait shapes):
pyramid, localized keypoints, filterKeypoints and Harris response against the oracle.  Needs a GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from visualslam_amd import capi, synth
ctx=capi.Context(0)
for shape,noct in (((2160,3840),0),((1234,2050),5),((3000,500),5)):
    img=synth.frame_np(*shape,kind="checker")
    t=time.time(); got=ctx.pyramid(img,noct,1.6); t1=time.time()-t
    t=time.time(); want=oracle.Pyramid(img,noct or oracle.auto_num_octaves(*shape),1.6); t2=time.time()-t
    print(shape, got.n_octaves, want.n_octaves, "gpu %.1f ms cpu %.1f s"%(t1*1e3,t2)); sys.stdout.flush()
    ok=True
    for o in range(got.n_octaves):
        for l in range(6): ok &= (got.gauss(o,l)==want.gauss(o,l)).all()
        for l in range(5): ok &= (got.dog(o,l)==want.dog(o,l)).all()
        wk=want.keypoints(o,3); gk,n=got.keypoints(o,3)
        ok &= n==len(wk) and gk.tobytes()==wk.tobytes()
        wf=want.filter_keypoints(o,wk); gf,nf=got.filter_keypoints(o,gk)
        ok &= nf==len(wf) and gf.tobytes()==wf.tobytes()
        print("  octave",o,want.sizes[o],"kp",n,"oriented",nf,"ok",bool(ok)); sys.stdout.flush()
    R=ctx.harris_response(img); ok &= R.tobytes()==oracle.harris_response(img).tobytes()
    print(" all ok:",bool(ok))
    got.close(); want.close()
