import importlib, sys, numpy as np
sys.path.insert(0,'/root/repo')
venc=importlib.import_module('video-encoder_amd')
w,h=1920,1080
eng=venc.Engine(w,h,qp=26,n_slices=8,max_items=1,in_slots=2,out_slots=2)
for t in range(6):
    eng.upload(t%2,*venc.synth_frame(w,h,t))
    eng.encode_step([t%2],[t%2],t==0,t&15,[0]); eng.sync()
    if t:
        c=eng.debug(9); f=eng.debug(10)
        print('t',t,'cnt mean %.1f median %d p90 %d p99 %d max %d  >32: %.3f  >64: %.3f flagged %.3f'%(c.mean(),np.median(c),np.percentile(c,90),np.percentile(c,99),c.max(),(c>32).mean(),(c>64).mean(),f.mean()))
        if t==3:
            cc=c.reshape(68,120)
            print('rows with many:', np.argwhere(cc>64)[:20].tolist())
