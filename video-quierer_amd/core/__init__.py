"""Drop-in for the reference package ``core`` (reference src/core/)."""
