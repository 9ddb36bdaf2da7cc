"""Drop-in for the reference's ``1DTFIM/`` folder (module names kept)."""
