"""Stand-in for the package the reference imports its transformer layers from
(`from VQCPCB.transformer.transformer_custom import ...`,
priors/transformer.py:12-15).  That third-party package is absent from the
reference tree and un-pinned; the layers provided here are this repository's own
specification (oracle/prior_oracle.py), computed by gfx950 HIP kernels."""
