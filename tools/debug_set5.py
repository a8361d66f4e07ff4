import os, sys, json, numpy as np
from PIL import Image
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import lerf_pytorch_amd as L
from oracle import lerf_oracle as O, c_oracle as CO
DATA=os.path.join(ROOT,'tests/data/Set5')
for model in ('lerf-g','lerf-l'):
    eng=L.LerfEngine.shipped(model)
    luts=O.load_luts(os.path.join(ROOT,'lerf-pytorch_amd/assets/models',model), linear=(model=='lerf-l'))
    for scale in (2,3,4):
        for n in ('baby','butterfly'):
            lr=np.array(Image.open(os.path.join(DATA,'LR_bicubic/rrLR_X%.2f_%.2f'%(scale,scale),n+'.png')))
            ref=CO.sr_u8(lr,luts,scale,scale,linear=(model=='lerf-l'))
            cf,ch=CO.lut_stages(lr,luts,1 if model=='lerf-l' else 3)
            f,h=eng.stages(lr)
            for fused in (True,False):
                o=eng.sr(lr,scale,fused=fused)
                d=np.abs(o.astype(int)-ref.astype(int))
                ys,xs,cs=np.nonzero(d)
                print(model,scale,n,'fused' if fused else 'unfused','stages_ok',np.array_equal(f,cf),np.array_equal(h,ch),'mismatch',int((d!=0).sum()),'max',int(d.max()), 'where', list(zip(ys[:5],xs[:5],cs[:5])), o.shape)
