"""No source change: a variant build that differs from the product by its EXTRA_FLAGS only (mkvar.sh)."""
