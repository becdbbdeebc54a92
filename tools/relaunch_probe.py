import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pantas_amd.engine import Graph, Synth
syn = Synth(seed=3, n_genes=5500, n_lines=10_000_000, base_id=10**7)
g = Graph(text=syn.gfa).to_device(0)
for where in ("host", "device"):
    g.set_tokenizer(where)
    g.augment_reset()
    b = g.load_gaf_mem(syn.gaf)
    print(where, "launches after reset:", [round(g.launch(b).main_ms, 3) for _ in range(6)])
    g.augment_reset()
    print(where, "after another reset:", [round(g.launch(b).main_ms, 3) for _ in range(4)])
    time.sleep(0.5)
    print(where, "after 0.5 s idle:", [round(g.launch(b).main_ms, 3) for _ in range(4)])
    b.free()
