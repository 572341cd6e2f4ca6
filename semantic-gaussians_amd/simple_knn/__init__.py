"""Drop-in `simple_knn` package (reference: submodules/simple-knn)."""
