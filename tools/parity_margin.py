import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, datagen
from test_host_logic import NAMES, _fit, _maxrel
from hpfrec_amd import cython_loops_float as be
df,nU,nI = datagen.readme_counts(); Y,iu,ii = datagen.triplets(df)
g = np.load('tests/golden/c1_full.npz')
for its in (1,2,5,10,20):
    i,arrs,_ = _fit(be,Y,iu,ii,nU,nI,30,its)
    print('c1',its, ' '.join('%s=%.1e'%(n,_maxrel(arrs[n], g['it%d_%s'%(its,n)])) for n in NAMES))
df,nU,nI = datagen.mid_counts(); Y,iu,ii = datagen.triplets(df)
g = np.load('tests/golden/mid_full.npz')
for its in (1,5,10):
    i,arrs,_ = _fit(be,Y,iu,ii,nU,nI,50,its)
    print('mid',its, ' '.join('%s=%.1e'%(n,_maxrel(arrs[n][::10], g['it%d_%s_rows'%(its,n)])) for n in NAMES))
