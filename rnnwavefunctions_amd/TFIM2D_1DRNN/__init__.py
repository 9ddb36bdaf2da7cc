"""Drop-in for the reference's ``2DTFIM_1DRNN/`` folder (module names kept)."""
