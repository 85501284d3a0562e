"""Fixture generators for tests, benches and smoke(): synthetic frames / crops (``synth``) and deterministic weight sets with the
properties the parity tests need -- margin-controlled detector weights, BN-calibrated and margin-carrier CRNN weights, seeded
TrOCR weights (``weights``).  NOT part of the product path: nothing under ``vtd_amd`` imports this package except the
explicit synthetic-weights opt-in of ``TransformerRecognizer`` (``model_name='seeded:<n>'``)."""
