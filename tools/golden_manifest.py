"""The fixtures under tests/golden/ and the script that writes each one (no reference import here: tests/ reads this file).

    python tools/gen_golden.py        # build container only: rebuilds EVERY fixture below (it runs the two sibling scripts as well)
"""
FIXTURES = {
    'g1_prompts.json': 'tools/gen_golden.py',
    'g2_tiling.json': 'tools/gen_golden.py',
    'g3g4_shuffle_masks.npz': 'tools/gen_golden.py',
    'g5g6_vlm.npz': 'tools/gen_golden.py',
    'g6b_ragged.npz': 'tools/gen_golden.py',
    'g7_vla.npz': 'tools/gen_golden.py',
    'g7b_vla_trace.npz': 'tools/gen_golden.py',
    'g7c_integrators.npz': 'tools/gen_golden.py',
    'g7d_general_masks.npz': 'tools/gen_golden.py',
    'g8_sft_grads.npz': 'tools/gen_golden.py',
    'g10_flow_matching.npz': 'tools/gen_golden.py',
    'g10b_flow_matching_vlm.npz': 'tools/gen_golden.py',
    'g11_packed.npz': 'tools/gen_golden.py',
    'META.json': 'tools/gen_golden.py',
    'g8_adapter.npz': 'tools/gen_golden_adapter.py',
    'g9_vla_state_keys.json': 'tools/gen_golden_vla_keys.py',
    'g12_resize.npz': 'tools/gen_golden_resize.py',          # Pillow's outputs (the resampler is a third-party dependency of the reference: no reference import)
}
