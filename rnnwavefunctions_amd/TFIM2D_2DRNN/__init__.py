"""Drop-in for the reference's ``2DTFIM_2DRNN/`` folder (module names kept)."""
