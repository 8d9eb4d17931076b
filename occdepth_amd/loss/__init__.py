"""Training-step losses and metrics (mirror of occdepth/loss/): SURVEY 8(f) rows N1 and N4."""
