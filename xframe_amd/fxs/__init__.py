"""Host-side mirror of ``xframe.projects.fxs`` for the reconstruct (MTIP phasing) path."""
