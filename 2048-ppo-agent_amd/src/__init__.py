"""Host-side mirror of the reference's ``src`` package, backed by libg2048.so on MI355X.

Import surface kept from the reference: src.actions, src.env_definitions, src.runs, src.ppo, src.optim,
src.stats.  ``src.g2048`` is new: the ctypes binding and the device-resident rollout engine.
"""
