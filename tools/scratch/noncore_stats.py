import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cloops_amd import api
from cloops_amd.synth import synth_chrom, chrom_sizes
name, length, n = chrom_sizes(200000000)[0]
X, Y = synth_chrom(n, length, 3000)
ch = api.Chromosome(X, Y)
for eps, m, cut in ((7500, 30, 5000), (5000, 50, 0), (10000, 20, 5000), (2500, 100, 5000)):
    c = ch.neighbor_counts(eps, cut)
    inn = c >= 0
    nc = inn & (c < m)
    res = ch.cluster("v2", eps, m, cut)
    lab = res.labels
    print(eps, m, cut, "in", int(inn.sum()), "noncore", int(nc.sum()), "mean cnt of noncore %.1f" % c[nc].mean(),
          "border", int((nc & (lab >= 0)).sum()), "noise", int((nc & (lab < 0)).sum()), "K", res.n_clusters,
          "pct of noncore cnt: ", np.percentile(c[nc], [10, 50, 90, 99]))
