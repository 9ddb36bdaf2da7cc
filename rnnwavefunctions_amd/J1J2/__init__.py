"""Drop-in for the reference's ``J1J2/`` folder (module names kept)."""
