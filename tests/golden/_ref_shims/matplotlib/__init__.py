"""Stand-in for matplotlib (absent from the build image) so that the reference's step-04 script can be imported for golden
generation: every plotting call is a no-op.  Test infrastructure only."""
