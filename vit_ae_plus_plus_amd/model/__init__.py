"""Drop-in mirror of the reference's ``model`` package: same module paths, class
names, constructor signatures, state-dict keys and return tuples; arithmetic by libvitae_hip.so."""
