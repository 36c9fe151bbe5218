"""Drop-in mirror of the reference's ``utils`` package for the pre-training path."""
