import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import oracle_lib as O
from twilight_amd import api, synth
import twilight_amd as twl
twl.init([0])
M = synth.nucleotide_matrix()
for name, batch, pk in [("flen128", synth.make_level_batch(5, 4000, members=(1, 1), seed=9), dict(flen=128)),
                        ("xdrop40", synth.make_level_batch(4, 4000, members=(1, 1), seed=21, sub=0.75, indel=0.05), dict(xdrop=40)),
                        ("flen700", synth.make_level_batch(4, 6000, members=((1,6),(1,6)), seed=33, sub=0.12), dict(flen=700, xdrop=9000))]:
    p = twl.make_params(M, **pk)
    oa, on, oerr, ost = O.align_batch(O.make_params(M, **pk), batch, threads=8)
    for mt in (1024, 0):
        twl.set_knob(api.KNOB_MT_MAX_PAIRS, mt)
        aln, n, err = twl.align_batch(p, batch)
        st = twl.get_stats(0)
        print(name, "mt", mt, "err", err.tolist(), "cells", st.band_cells, "oracle", ost.cells, "spec", st.speculative, "relaunched", st.n_relaunched, flush=True)
