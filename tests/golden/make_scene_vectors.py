#!/usr/bin/env python3
"""Generate tests/golden/scene_spots.json: the CPU oracle's spot lists for a dozen deterministic
synthetic scenes (tests/scenes.py).  The JSON holds seeds and expected outputs only; the IQ is
regenerated from the seed (numpy PCG64).  It pins the oracle against accidental edits and lets the
GPU tests check the product without the oracle in the loop."""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol
import scenes

out = []
for seed in scenes.GOLDEN_SEEDS:
    I, Q = scenes.make_scene(seed)
    spots, _, _ = ol.decode(I, Q, scenes.NS)
    out.append({"seed": seed, "spots": [scenes.spot_record(s) for s in spots]})
json.dump({"scenes": out}, open(os.path.join(HERE, "scene_spots.json"), "w"), indent=1)
print("wrote", sum(len(s["spots"]) for s in out), "spots for", len(out), "scenes")
