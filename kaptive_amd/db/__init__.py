"""Database container consumed by the typing path (reference: src/kaptive/db/)."""

from kaptive_amd.db.core import Database
from kaptive_amd.db.models import DatabaseError, DatabaseMetadata, Phenotype, Phenotypes

__all__ = ["Database", "DatabaseError", "DatabaseMetadata", "Phenotype", "Phenotypes"]
