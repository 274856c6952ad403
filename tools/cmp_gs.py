import numpy as np
a=np.load('gpurun_out/r3i/on.npz'); b=np.load('gpurun_out/r3i/off.npz')
for k in ('l1','l2','l3','rec'):
    d=np.abs(a[k]-b[k]); i=np.unravel_index(d.argmax(), d.shape)
    print(k, a[k].shape, 'max diff', d.max(), 'at', i, 'vals', a[k][i], b[k][i], 'scale', np.abs(b[k]).max(), 'n>0.05*scale', int((d>0.05*np.abs(b[k]).max()).sum()))
