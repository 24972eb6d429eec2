"""`make_prg update` host side: denovo_paths.txt parsing and maximum-likelihood paths (reference make_prg/update/)."""
