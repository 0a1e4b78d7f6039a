import sys, warnings
sys.path.insert(0, '.')
import numpy as np
from kiez_amd import Kiez
from oracle import kiez_oracle as O
warnings.simplefilter('ignore')
rng = np.random.RandomState(600)
s = rng.rand(300, 300).astype(np.float32); t = rng.rand(5000, 300).astype(np.float32)
kz = Kiez(n_candidates=10, algorithm='SklearnNN', algorithm_kwargs=dict(metric='euclidean'), hubness='MutualProximity', hubness_kwargs={'method':'normal'})
kz.fit(s, t)
nn = kz.algorithm
d_s2t, i_s2t = nn.kneighbors(k=10)
d_t2s, i_t2s = nn.kneighbors(k=10, query=t, s_to_t=False)
od, oi, inter = O.kiez_pipeline(s, t, 10, 5, 'euclidean', 2, 'MutualProximity', {'method':'normal'}, return_intermediates=True)
print('s2t idx eq', np.array_equal(i_s2t, inter['ind_s2t']), 'max rel', np.abs(d_s2t-inter['dist_s2t']).max()/6, 'n diff', (d_s2t!=inter['dist_s2t']).sum())
print('t2s idx eq', np.array_equal(i_t2s, inter['ind_t2s']), 'n diff', (d_t2s!=inter['dist_t2s']).sum(), 'of', d_t2s.size)
tr, _ = kz.hubness.transform(inter['dist_s2t'], inter['ind_s2t'], s)
print('transform on oracle inputs: max abs', np.abs(tr-inter['transformed']).max(), 'max rel', (np.abs(tr-inter['transformed'])/np.abs(inter['transformed'])).max())
mu = kz.hubness.mu_t_to_s_.numpy(); sd = kz.hubness.sd_t_to_s_.numpy()
print('mu diff', np.abs(mu-np.nanmean(inter['dist_t2s'],axis=1)).max(), 'sd rel diff', (np.abs(sd-np.nanstd(inter['dist_t2s'],axis=1))/sd).max())
bad = np.argwhere(d_t2s!=inter['dist_t2s'])
print(bad[:5]); 
for r,c in bad[:5]: print(repr(d_t2s[r,c]), repr(inter['dist_t2s'][r,c]))
