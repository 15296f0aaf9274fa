"""Only the correlation layer of the reference's PWCNet package is on the hot path."""
