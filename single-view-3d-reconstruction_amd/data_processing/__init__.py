"""Host-side mirror of the reference's data_processing package, restricted to what sits next to the hot path
(SURVEY.md section 8 rows f3 / f4): on-device occupancy labelling (libmesh.inside_mesh, implicit_waterproofing,
mesh_occupancies) and the sample wire formats (volume_reader, sample_io)."""
