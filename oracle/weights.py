"""The deterministic key-named parameter fill lives in stylish_tts_amd/synthetic_weights.py (bench.py builds its
random-init models with it without touching oracle/); re-exported here for the tests, tools and fixtures."""
from stylish_tts_amd.synthetic_weights import (_fan_in, _power_iteration, _rng, fill_state_dict,  # noqa: F401
                                               fill_tensor)
