"""Drop-in for the reference package ``indexes`` (reference src/indexes/)."""
